// traj_space_mfma.hip -- fused space step of trajectory attention (attention.py:524-535), bf16, head dim 64.
//
// For every query s and every frame f:  x~[s,f,:] = softmax_p(scale * q_s . k_{f,p}) . v_{f,p}   (P keys per frame)
// The reference materialises the S x S logits (118 MB per clip and block); here a workgroup owns 128 queries of
// one (batch, head) and walks the F frames: the frame's K and V^T tiles sit in LDS, each wave (32 queries) computes
// the 32-key x 32-query logit tiles with v_mfma_f32_32x32x16_bf16 in the SWAPPED orientation (keys on the
// accumulator rows, queries on the lanes), so the whole per-(query,frame) softmax is lane-local (one exchange with
// the partner half-wave), and the normalised probabilities feed the P.V product straight from the accumulator
// registers as the B operand (cdna_hip_programming.md section 3, "accumulator tile as the next MFMA's operand").
// P <= 224: the frame's keys fit the register tile and the softmax is exact in one pass.  224 < P <= 448 (the HR
// 16x336 config: 21x21 patches + objects): the frame is cut into NT key tiles of NKB*32 keys (NKB*NT == ceil(P/32),
// so only the last block of the last tile is ragged) that are merged by an online softmax (one rescale of the 32
// output accumulators per extra tile; frames never share a tile, attention.py:524-529).  Outputs x~ [B,S,F,C],
// x_diag [B,S,C] (= x~ at the query's own frame) and the per-(query,frame) log-sum-exp for backward; rows leave
// through a per-wave LDS slab so every global store is a 16-byte piece of a 128-byte row.
#include "focus_common.h"
#include "traj_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int HD = 64;                 // head dim
constexpr int QT = 128;                // queries per workgroup (4 waves x 32)

// 16-B chunk swizzle of a [rows][64 bf16] tile with 128-B rows.  A row covers half of the 64 LDS banks (row parity
// picks the half), so the XOR key is built from row>>1, bit-reversed: (a) a ds_read_b128 bank group reads one chunk
// of 16 rows whose row>>1 values are 8 distinct ones mod 8 -> 16 distinct (half, chunk) slots; (b) a
// ds_read_b64_tr_b16 bank group reads 4 adjacent chunks of rows r..r+3 -> rows r and r+2 need keys that differ in
// bit 2, which the reversal provides.  (key = row & 7 was 2-way conflicted on both: SQ_LDS_BANK_CONFLICT ~40 %.)
__device__ __forceinline__ int swz(int row, int chunk) {
    const int key = ((row & 2) << 1) | ((row >> 1) & 2) | ((row >> 3) & 1);
    return row * 128 + ((chunk ^ key) << 4);
}

union Pack8 { bf16x8 v; bf16_t e[8]; uint4 u; s16x4 t[2]; };

// V^T[d][key] fragment (A operand of  y^T = V^T . P) gathered from the ROW-major V tile with two transposed LDS reads:
// element j <-> key r0 + 8*(j>>2) + (j&3), d = c0 + (lane & 31); the caller passes r0 = 16*s + 4*(lane>>5) (+ block).
__device__ __forceinline__ bf16x8 col_frag(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    const int row = r0 + (i >> 2);
    Pack8 p;
    p.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + swz(row, col >> 3) + (col & 4) * 2));
    p.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + swz(row + 8, col >> 3) + (col & 4) * 2));
    return p.v;
}

template <int NKB, bool MULTI>
__global__ __launch_bounds__(256, 2) void traj_space_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ xt,
                                                                bf16_t* __restrict__ xdiag, float* __restrict__ lse,
                                                                int B, int F, int P, int heads, int NTarg) {
    constexpr int KROWS = NKB * 32;
    // K and V tiles are SEPARATE LDS objects on purpose: with one array the compiler cannot tell a ds_read of K from
    // an LDS-DMA to V in flight and drains vmcnt(0) before every first read (cdna_hip_programming.md section 5)
    __shared__ __attribute__((aligned(1024))) char sK[KROWS * 128];   // [KROWS][64] bf16, 128-B rows, chunk-swizzled
    __shared__ __attribute__((aligned(1024))) char sV[KROWS * 128];   // same layout for V (read by columns: col_frag)
    __shared__ __attribute__((aligned(1024))) char slabs[4 * 4096];   // 4 x [32][128 B] output slabs, one per wave

    const int S = F * P, N = S + 1, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    int bx, bh;
    focus_xcd_group(bx, bh);                                   // the query tiles of one (b, h) share an XCD's L2
    const int b = bh / heads, hh = bh % heads;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int s_q = min(bx * QT + w * 32 + r, S - 1);  // this lane's query (clamped)
    const bool q_valid = bx * QT + w * 32 + r < S;
    const bf16_t* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float c2 = rsqrtf((float)HD) * 1.44269504088896341f;   // scale * log2(e)
    const int NT = MULTI ? NTarg : 1;                            // key tiles per frame
    const int U = F * NT;                                        // stream steps: (frame, key tile)

    // Q fragments: B operand of the swapped product, lane (r,h) holds Q[q=r][16*ks + 8h + j]
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)(1 + s_q) * tok + ks * 16 + 8 * h);

    char* slab = slabs + w * 4096;
    // ---- K_f / V_f staging by LDS-DMA (global_load_lds_dwordx4), issued by the same 4 waves and overlapped by phase:
    // K_{f+1} streams in while frame f's softmax and P.V run, V_{f+1} while frame f's rows leave and frame f+1's
    // logits are computed.  One wave instruction fills 8 tile rows (1 KiB, lane-linear); the chunk swizzle is
    // applied on the SOURCE address; rows >= P are clamped to row P-1 (their logits are masked below).
    // Waits are counted: a wave's only other vector-memory traffic is stores, and loads retire in order, so
    // "at most NKB outstanding" means every load older than the last NKB has landed.
    const int drow = lane >> 3, dkey = ((drow & 2) << 1) | ((drow >> 1) & 2);          // key bits of row&7 (see swz)
    auto dma_tile = [&](char* tile, int f, int kbase, int part) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NKB; ++g) {
            const int t = g * 4 + w;                                  // 8-row group of the tile
            const int row = t * 8 + drow;
            const int key = dkey | ((t & 1));                         // bit 3 of the row = bit 0 of t
            const bf16_t* src = base + (int64_t)(1 + f * P + min(kbase + row, P - 1)) * tok + part * C + (((lane & 7) ^ key) << 3);
            glds16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(tile) + t * 1024));
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q loads retired in the compiler's bookkeeping too
    dma_tile(sK, 0, 0, 1);
    dma_tile(sV, 0, 0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // online-softmax state of the current frame (MULTI): running max (raw logits), this lane's partial sum, outputs
    float m_run = -INFINITY, sum_run = 0.f;
    f32x16 y[2];
    int f = 0, j = 0;                                 // frame and key tile of step u
    for (int u = 0; u < U; ++u) {
        const bool more = u + 1 < U;
        const int jn = (MULTI && j + 1 < NT) ? j + 1 : 0, fn = jn ? f : f + 1;   // step u+1
        const int kbase = MULTI ? j * KROWS : 0;
        const bool first = !MULTI || j == 0, last = !MULTI || j == NT - 1;
        // ---- logits: acc[kb][reg] = sum_d K[kb*32 + row(reg,h)][d] * Q[q=r][d] ----
        f32x16 acc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + swz(kb * 32 + r, ks * 2 + h));
                acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc[kb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // A: every wave has read K_f
        if (more) dma_tile(sK, fn, jn * KROWS, 1);
        // ---- per-(query, frame) softmax over the P keys: this lane holds half of its query's keys.
        // VALU-lean form (the kernel is softmax-bound, not MFMA-bound): raw-logit max, one fma + one exp2 per
        // element, only the tail key block is masked, and the 1/sum normalisation is applied to the 32 outputs
        // after P.V instead of to the 16*NKB probabilities. ----
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            // NKB == ceil(P/32) (one instantiation per block count): only the LAST block holds padded keys, and that
            // is known at compile time -- a run-time test per block made the compiler materialise all 16*NKB key
            // masks (cmp + s_or + cndmask + SGPR spills: ~900 of the ~2000 instructions per frame)
            if (kb == NKB - 1) {
                const int lim = P - kbase - kb * 32 - 4 * h;   // row (i&3) + 8*(i>>2) of this lane's half is real iff < lim
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) >= lim) acc[kb][i] = -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, acc[kb][i]);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float alpha = 1.f;                            // rescale of what earlier key tiles of this frame accumulated
        if (MULTI) {
            if (first) { m_run = m; sum_run = 0.f; }
            else {
                const float mn = fmaxf(m_run, m);
                alpha = __builtin_amdgcn_exp2f((m_run - mn) * c2);
                m_run = mn; m = mn;
            }
        }
        const float m2 = m * c2;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pexp = __builtin_amdgcn_exp2f(fmaf(acc[kb][i], c2, -m2));
                acc[kb][i] = pexp;
                sum += pexp;
            }
        if (MULTI) { sum_run = fmaf(sum_run, alpha, sum); sum = sum_run; }
        float inv = 0.f;
        if (last) {
            sum += __shfl_xor(sum, 32, 64);
            inv = 1.f / sum;
            if (h == 0 && q_valid)
                lse[(((int64_t)b * heads + hh) * S + s_q) * F + f] = (m2 + __builtin_amdgcn_logf(sum)) * 0.69314718055994531f;
        }

        // V_f was issued before the K_{f+1} pieces above: at most NKB loads outstanding <=> V_f has landed
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // D: V_f is visible to every wave
        // ---- y[dblk][reg] = sum_key V^T[d][key] * P[key][q]  (P straight from the accumulators) ----
        if (first) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { y[0][i] = 0.f; y[1][i] = 0.f; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { y[0][i] *= alpha; y[1][i] *= alpha; }
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                Pack8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf.e[j] = f32_to_bf16(acc[kb][8 * s2 + j]);   // un-normalised, <= 1
                const int k0 = kb * 32 + 16 * s2 + 4 * h;
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) {
                    const bf16x8 vf = col_frag(sV, k0, dblk * 32, lane);
                    y[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, y[dblk], 0, 0, 0);
                }
            }

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // B: every wave has read V_f
        if (more) {
            dma_tile(sV, fn, jn * KROWS, 2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKB) : "memory");   // K_{u+1} (issued after A) has landed
            __builtin_amdgcn_s_barrier();             // C: K_{u+1} is visible to every wave
        }
        if (!last) { f = fn; j = jn; continue; }      // (MULTI) the frame's next key tile; rows leave after its last one
        // ---- rows out through the wave's LDS slab: [32 q][16 chunks of 8 B], chunk ^= q & 15 ----
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(y[dblk][4 * g + 0] * inv) | ((uint32_t)f32_to_bf16(y[dblk][4 * g + 1] * inv) << 16);
                pk.y = (uint32_t)f32_to_bf16(y[dblk][4 * g + 2] * inv) | ((uint32_t)f32_to_bf16(y[dblk][4 * g + 3] * inv) << 16);
                *reinterpret_cast<uint2*>(slab + r * 128 + (((dblk * 8 + 2 * g + h) ^ (r & 15)) << 3)) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave re-reads only its own slab
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
            const int row = p4 * 8 + (lane >> 3), q8 = lane & 7;
            uint4 raw = *reinterpret_cast<const uint4*>(slab + row * 128 + ((q8 ^ ((row & 15) >> 1)) << 4));
            if (row & 1) { const uint32_t a0 = raw.x, a1 = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = a0; raw.w = a1; }
            const int s_row = bx * QT + w * 32 + row;
            if (s_row < S) {
                *reinterpret_cast<uint4*>(xt + (((int64_t)b * S + s_row) * F + f) * C + hh * HD + q8 * 8) = raw;
                if (s_row / P == f)
                    *reinterpret_cast<uint4*>(xdiag + ((int64_t)b * S + s_row) * C + hh * HD + q8 * 8) = raw;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next frame overwrites it
        f = fn; j = jn;
    }
}

template <int NKB, bool MULTI>
int launch_fwd(const void* qkv, void* xt, void* xdiag, float* lse, int B, int F, int P, int heads, int NT, hipStream_t s) {
    const int S = F * P;
    dim3 grid((S + QT - 1) / QT, B * heads);
    hipLaunchKernelGGL((traj_space_fwd_kernel<NKB, MULTI>), grid, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)xt,
                       (bf16_t*)xdiag, lse, B, F, P, heads, NT);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

// Key tiling of one frame: NB = ceil(P/32) blocks of 32 keys = NT tiles of NKB blocks, NKB <= 7 the largest divisor of
// NB (so every tile is full except the last block of the last tile).  P <= 224 -> one tile.
void focus_traj_space_tiling(int P, int* nkb, int* nt) {
    const int NB = (P + 31) / 32;
    int k = NB < 7 ? NB : 7;
    while (NB % k) --k;
    *nkb = k; *nt = NB / k;
}

bool focus_traj_space_mfma_ok(int P, int d, int heads, int dtype) {
    static const bool enabled = !(getenv("FOCUS_TRAJ_FUSED") && atoi(getenv("FOCUS_TRAJ_FUSED")) == 0);
    return enabled && dtype == FOCUS_BF16 && d == HD && P >= 1 && P <= 32 * FOCUS_TRAJ_MAX_KEY_BLOCKS &&
           ((heads * HD) % 8) == 0;
}

// Patch-token rows only (xt, xdiag, lse); the cls row is handled by the caller.
int focus_traj_space_fwd_mfma(const void* qkv, void* xt, void* xdiag, float* lse, int B, int F, int P, int heads,
                              hipStream_t s) {
    if (B * heads > 65535) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(qkv, 16) || !focus_aligned(xt, 16) || !focus_aligned(xdiag, 16)) return FOCUS_ERR_ALIGN;
    int nkb, nt;
    focus_traj_space_tiling(P, &nkb, &nt);
#define FWD(K) case K: return nt == 1 ? launch_fwd<K, false>(qkv, xt, xdiag, lse, B, F, P, heads, 1, s) \
                                      : launch_fwd<K, true>(qkv, xt, xdiag, lse, B, F, P, heads, nt, s)
    switch (nkb) {               // exact block count: the kernels rely on NKB * NT == ceil(P/32)
        FWD(1); FWD(2); FWD(3); FWD(4); FWD(5); FWD(6); FWD(7);
        default: break;
    }
#undef FWD
    return FOCUS_ERR_SHAPE;
}
