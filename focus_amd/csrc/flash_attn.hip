// flash_attn.hip -- multi-head attention without the [Nq, Nk] probabilities in memory (bf16, head dim 32 / 48 / 64).
//
// The STEVE decoder's causal self-attention over the 1024 image tokens of a frame (STEVE/transformer.py:23-49 with the
// upper-triangular mask of :131-132, :149-151; steve.py:303-306): 4 heads of 48 channels, B*T = 768 sequences at the
// BASELINE shape.  The reference materialises softmax(q k^T) -- 6.4 GB per block in bf16 -- and, in training, a dropout
// mask of the same size (transformer.py:44-45).  Here:
//   out = dropout(softmax(scale * q k^T + mask)) v        lse[b,h,q] = log sum_k exp(scale * q.k)      (forward)
//   dV = Pd^T dO,  dP = dO V^T,  dS = P * (dP * keep/(1-p) - delta) * scale,  dQ = dS K,  dK = dS^T Q     (backward)
// with P recomputed from lse, Pd = P * keep / (1 - p) and delta[q] = dO[q,:] . out[q,:].  The dropout mask is a function of
// (seed, batch*head, query, key) evaluated where it is needed (drop_keep below), so the three kernels agree on it and
// nothing of size Nq x Nk is ever stored; p is quantised to 16 bits (0.1 -> 6554 / 65536).
//
// Kernel structure = the space step of trajectory attention (traj_space_mfma.hip, traj_space_bwd_mfma.hip): swapped
// v_mfma_f32_32x32x16_bf16 products (keys on the accumulator rows, queries on the lanes) so the softmax of a query is
// lane-local, probabilities feed the next product straight from the accumulator registers, K / V tiles reach LDS by LDS-DMA
// into 128-byte rows (a 48-channel row uses 96 of them; the two padding chunks hold a copy of the last real chunk and only
// ever meet output rows / columns that are not stored), transposed operands come from ds_read_b64_tr_b16.
//   flash_fwd_kernel : workgroup = 128 queries of one (b, h), 4 waves x 32; walks key tiles of NKB x 32 keys with an online
//                      softmax; causal: tiles past the workgroup's last query are not visited, blocks past a wave's last
//                      query are skipped, only blocks on the diagonal are masked element-wise; heavy query tiles first.
//   flash_dq_kernel  : same ownership; also forms delta from its own dO / out rows and writes it for the dK/dV kernel.
//   flash_dkv_kernel : workgroup = 128 keys of one (b, h), one wave per 32 keys (K, V fragments in registers), walks the
//                      32-query chunks at or below the diagonal through a 4-stage LDS-DMA ring of (Q | dO | lse, delta).
// No atomics; every output is written exactly once; the backward is deterministic.
#include "focus_common.h"
#include "traj_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int QT = 128;                 // queries per workgroup (forward, dQ) / keys per workgroup (dK, dV)
constexpr float LOG2E = 1.44269504088896341f;

union Pack8 { bf16x8 v; bf16_t e[8]; uint4 u; uint2 h2[2]; s16x4 t[2]; };

// 16-byte chunk swizzle of a [rows][128 B] tile (see traj_space_mfma.hip: conflict-free for ds_read_b128 of 16 rows and
// for ds_read_b64_tr_b16 of 4 adjacent chunks of rows r..r+3)
__device__ __forceinline__ int swz(int row, int chunk) {
    const int key = ((row & 2) << 1) | ((row >> 1) & 2) | ((row >> 3) & 1);
    return row * 128 + ((chunk ^ key) << 4);
}

// MFMA 32x32x16 operand whose k index runs over tile ROWS and whose m/n index is a tile COLUMN, in the k order of an
// accumulator tile used as the other operand: element j <-> row r0 + 8*(j>>2) + (j&3), column c0 + (lane&31); the caller
// passes r0 = 16*s + 4*(lane>>5) (+ block base)
__device__ __forceinline__ bf16x8 col_frag(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    const int row = r0 + (i >> 2);
    Pack8 p;
    p.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + swz(row, col >> 3) + (col & 4) * 2));
    p.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + swz(row + 8, col >> 3) + (col & 4) * 2));
    return p.v;
}

__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s2) {
    Pack8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p.e[j] = f32_to_bf16(a[8 * s2 + j]);
    return p.v;
}

// The dropout decision of element (query q, key k) of head bh: one 32-bit mix (lowbias32) of (seed, bh, q, k >> 1) serves the
// key pair (k & ~1, k | 1) with its low / high half; keep <=> half >= thr (thr = round(p * 65536)).  oracle:
// tests/test_gpu_flash.py drop_keep_reference is the same arithmetic in numpy.
__device__ __forceinline__ uint32_t drop_mix(uint32_t base, uint32_t q, uint32_t kpair) {
    uint32_t x = base ^ (q * 0x85EBCA77u) ^ (kpair * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_base(uint32_t seed, uint32_t bh) { return seed ^ (bh * 0x9E3779B1u); }

struct Geo {                                   // what every kernel derives from the argument block
    const bf16_t* q; const bf16_t* k; const bf16_t* v;
    int64_t ldq, ldk, ldv;
    int Nq, Nk, heads, b, hh;
    float scale, c2, keep_scale;
    uint32_t thr, dbase;
    bool causal;
};

template <int D>
__device__ __forceinline__ Geo geo_of(const focus_flash_args& a, int bh) {
    Geo g;
    g.Nq = a.Nq; g.Nk = a.Nk; g.heads = a.heads;
    g.b = bh / a.heads; g.hh = bh % a.heads;
    g.q = static_cast<const bf16_t*>(a.q) + g.b * a.bsq + g.hh * D;
    g.k = static_cast<const bf16_t*>(a.k) + g.b * a.bsk + g.hh * D;
    g.v = static_cast<const bf16_t*>(a.v) + g.b * a.bsv + g.hh * D;
    g.ldq = a.ldq; g.ldk = a.ldk; g.ldv = a.ldv;
    g.scale = a.scale; g.c2 = a.scale * LOG2E;
    g.thr = a.drop_thr;
    g.keep_scale = a.drop_thr ? 65536.f / (float)(65536u - a.drop_thr) : 1.f;
    g.dbase = a.drop_thr ? drop_base(*a.seed, (uint32_t)bh) : 0u;
    g.causal = a.causal != 0;
    return g;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int D, int NKB>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(const focus_flash_args a) {
    constexpr int KS = D / 16, CH = D / 8, KROWS = NKB * 32;
    // K and V tiles are separate LDS objects (one array would make the compiler drain vmcnt(0) before every first read)
    __shared__ __attribute__((aligned(1024))) char sK[KROWS * 128];
    __shared__ __attribute__((aligned(1024))) char sV[KROWS * 128];
    __shared__ __attribute__((aligned(1024))) char slabs[4 * 4096];

    int bxr, bh;
    focus_xcd_group(bxr, bh);
    const int bx = gridDim.x - 1 - bxr;                        // causal: the query tiles with the most keys start first
    const Geo g = geo_of<D>(a, bh);
    const int Nq = g.Nq, Nk = g.Nk;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = bx * QT, qmin_w = q0 + w * 32;
    const int s_q = min(qmin_w + r, Nq - 1);
    const int qmax_w = min(qmin_w + 31, Nq - 1);
    const int ntiles = g.causal ? min((Nk + KROWS - 1) / KROWS, min(q0 + QT - 1, Nq - 1) / KROWS + 1) : (Nk + KROWS - 1) / KROWS;

    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(g.q + (int64_t)s_q * g.ldq + ks * 16 + 8 * h);

    char* slab = slabs + w * 4096;
    const int drow = lane >> 3, dkey = ((drow & 2) << 1) | ((drow >> 1) & 2);
    auto dma_tile = [&](char* tile, int kt, const bf16_t* src_base, int64_t ld) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < NKB; ++gq) {
            const int t = gq * 4 + w;                                  // 8-row group of the tile
            const int row = t * 8 + drow;
            const int chunk = min((lane & 7) ^ (dkey | (t & 1)), CH - 1);   // padding chunks: a copy of the last real one
            const bf16_t* src = src_base + (int64_t)min(kt * KROWS + row, Nk - 1) * ld + chunk * 8;
            glds16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(tile) + t * 1024));
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q loads retired in the compiler's bookkeeping too
    dma_tile(sK, 0, g.k, g.ldk);
    dma_tile(sV, 0, g.v, g.ldv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float m_run = -INFINITY, sum_run = 0.f;
    f32x16 y[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { y[0][i] = 0.f; y[1][i] = 0.f; }
    for (int u = 0; u < ntiles; ++u) {
        const bool more = u + 1 < ntiles;
        // ---- logits: acc[kb][reg] = sum_d K[key][d] * Q[q = r][d]; blocks wholly past this wave's last query are dead ----
        f32x16 acc[NKB];
        bool live[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int blk0 = u * KROWS + kb * 32;
            live[kb] = blk0 < Nk && (!g.causal || blk0 <= qmax_w);
            if (live[kb]) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[kb][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + swz(kb * 32 + r, ks * 2 + h));
                    acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc[kb], 0, 0, 0);
                }
                // element-wise masks only where the block meets the diagonal or the end of the keys (wave-uniform test)
                if ((g.causal && blk0 + 31 > qmin_w) || blk0 + 32 > Nk) {
                    const int lim = (g.causal ? min(s_q + 1, Nk) : Nk) - blk0 - 4 * h;   // row (i&3) + 8*(i>>2) is real iff < lim
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((i & 3) + 8 * (i >> 2) >= lim) acc[kb][i] = -INFINITY;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[kb][i] = -INFINITY;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // A: every wave has read K_u
        if (more) dma_tile(sK, u + 1, g.k, g.ldk);
        // ---- online softmax: this lane holds half of the tile's keys of its query ----
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, acc[kb][i]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float alpha = 1.f;
        if (u == 0) { m_run = m; }                    // (tile 0 holds key 0 <= every query: m is finite)
        else {
            const float mn = fmaxf(m_run, m);
            alpha = __builtin_amdgcn_exp2f((m_run - mn) * g.c2);
            m_run = mn;
        }
        const float m2 = m_run * g.c2;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pexp = __builtin_amdgcn_exp2f(fmaf(acc[kb][i], g.c2, -m2));
                acc[kb][i] = pexp;
                sum += pexp;
            }
        sum_run = fmaf(sum_run, alpha, sum);
        if (g.thr) {                                   // dropout acts on the probabilities that meet V, not on the sum
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (!live[kb]) continue;
                const uint32_t kp0 = (uint32_t)(u * KROWS + kb * 32 + 4 * h) >> 1;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const uint32_t x = drop_mix(g.dbase, (uint32_t)s_q, kp0 + (uint32_t)(((i & 3) + 8 * (i >> 2)) >> 1));
                    if ((x & 0xffffu) < g.thr) acc[kb][i] = 0.f;
                    if ((x >> 16) < g.thr) acc[kb][i + 1] = 0.f;
                }
            }
        }
        // V_u was issued before the K_{u+1} pieces above: at most NKB loads outstanding <=> V_u has landed
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // D: V_u is visible to every wave
        if (u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { y[0][i] *= alpha; y[1][i] *= alpha; }
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (!live[kb]) continue;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack_acc(acc[kb], s2);               // un-normalised, <= 1
                const int k0 = kb * 32 + 16 * s2 + 4 * h;
#pragma unroll
                for (int dblk = 0; dblk < (D + 31) / 32; ++dblk) {
                    const bf16x8 vf = col_frag(sV, k0, dblk * 32, lane);
                    y[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, y[dblk], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // B: every wave has read V_u
        if (more) {
            dma_tile(sV, u + 1, g.v, g.ldv);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKB) : "memory");   // K_{u+1} (issued after A) has landed
            __builtin_amdgcn_s_barrier();             // C: K_{u+1} is visible to every wave
        }
    }
    float sum = sum_run + __shfl_xor(sum_run, 32, 64);
    const float inv = g.keep_scale / sum;
    const bool q_valid = qmin_w + r < Nq;
    if (h == 0 && q_valid)
        a.lse[((int64_t)g.b * g.heads + g.hh) * Nq + s_q] = (m_run * g.c2 + __builtin_amdgcn_logf(sum)) * 0.69314718055994531f;
    // ---- rows out through the wave's LDS slab: [32 q][16 chunks of 8 B], chunk ^= q & 15 ----
#pragma unroll
    for (int dblk = 0; dblk < (D + 31) / 32; ++dblk)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(y[dblk][4 * gq + 0] * inv) | ((uint32_t)f32_to_bf16(y[dblk][4 * gq + 1] * inv) << 16);
            pk.y = (uint32_t)f32_to_bf16(y[dblk][4 * gq + 2] * inv) | ((uint32_t)f32_to_bf16(y[dblk][4 * gq + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(slab + r * 128 + (((dblk * 8 + 2 * gq + h) ^ (r & 15)) << 3)) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* out = static_cast<bf16_t*>(a.out) + g.b * a.bso + g.hh * D;
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
        const int row = p4 * 8 + (lane >> 3), q8 = lane & 7;
        uint4 raw = *reinterpret_cast<const uint4*>(slab + row * 128 + ((q8 ^ ((row & 15) >> 1)) << 4));
        if (row & 1) { const uint32_t a0 = raw.x, a1 = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = a0; raw.w = a1; }
        const int s_row = qmin_w + row;
        if (s_row < Nq && q8 < CH) *reinterpret_cast<uint4*>(out + (int64_t)s_row * a.ldo + q8 * 8) = raw;
    }
}

// ------------------------------------------------------------------------------------------------
// dQ (and delta).  Stream step t = 32-key block t: ring stage t & 3 = [K block 32 x 128 B | V block 32 x 128 B]; wave w
// DMAs rows 8w..8w+7 of both (2 instructions per step), three steps ahead.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void flash_dq_kernel(const focus_flash_args a) {
    constexpr int KS = D / 16, CH = D / 8;
    __shared__ __attribute__((aligned(1024))) char ring[4 * 8192];       // after the loop: the 4 output slabs

    int bxr, bh;
    focus_xcd_group(bxr, bh);
    const int bx = gridDim.x - 1 - bxr;
    const Geo g = geo_of<D>(a, bh);
    const int Nq = g.Nq, Nk = g.Nk;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = bx * QT, qmin_w = q0 + w * 32;
    const bool q_valid = qmin_w + r < Nq;
    const int s_q = min(qmin_w + r, Nq - 1);
    const int qmax_w = min(qmin_w + 31, Nq - 1);
    const int nblk = (Nk + 31) / 32;
    const int T = g.causal ? min(nblk, min(q0 + QT - 1, Nq - 1) / 32 + 1) : nblk;

    const int drow = lane >> 3, dkey = ((drow & 2) << 1) | ((drow >> 1) & 2);
    const int dchunk = min((lane & 7) ^ (dkey | (w & 1)), CH - 1);
    const uint32_t ring_a = lds_addr_of(ring);
    auto dma_step = [&](int t) __attribute__((always_inline)) {          // K and V rows 8w..8w+7 of block t
        const int64_t row = min(t * 32 + w * 8 + drow, Nk - 1);          // padded keys: a copy of the last real row
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_a + (t & 3) * 8192 + w * 1024);
        glds16(g.k + row * g.ldk + dchunk * 8, dst);
        glds16(g.v + row * g.ldv + dchunk * 8, dst + 4096);
    };

    // ---- prologue: Q and dO fragments (B operands: [k = d][col = q]), delta, softmax statistics ----
    const bf16_t* dob = static_cast<const bf16_t*>(a.dout) + g.b * a.bsdo + g.hh * D;
    const bf16_t* ob = static_cast<const bf16_t*>(a.out) + g.b * a.bso + g.hh * D;
    bf16x8 qf[KS], df[KS];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(g.q + (int64_t)s_q * g.ldq + ks * 16 + 8 * h);
        Pack8 da, oa;
        da.u = *reinterpret_cast<const uint4*>(dob + (int64_t)s_q * a.lddo + ks * 16 + 8 * h);
        oa.u = *reinterpret_cast<const uint4*>(ob + (int64_t)s_q * a.ldo + ks * 16 + 8 * h);
        df[ks] = da.v;
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum = fmaf(bf16_to_f32(da.e[j]), bf16_to_f32(oa.e[j]), dsum);
    }
    const int64_t stat = ((int64_t)g.b * g.heads + g.hh) * Nq + s_q;
    const float lse2 = q_valid ? a.lse[stat] * LOG2E : INFINITY;          // +inf -> P = 0 for padded queries
    dsum += __shfl_xor(dsum, 32, 64);
    const float dels = dsum * g.scale;
    if (h == 0 && q_valid) a.delta[stat] = dels;                          // (already times scale) for the dK / dV kernel
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ordinary loads above are done before any DMA is counted
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(qf[ks])); asm volatile("" : "+v"(df[ks])); }
    dma_step(0);
    if (T > 1) dma_step(1);
    if (T > 2) dma_step(2);
    const float ks_scale = g.keep_scale * g.scale;

    for (int t = 0; t < T; ++t) {
        // step t landed?  issued after it: steps t+1, t+2 (2 instructions each)
        if (t + 2 < T) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // step t visible to all; everyone is done with stage (t-1) & 3
        if (t + 3 < T) dma_step(t + 3);
        const int blk0 = t * 32;
        if (g.causal && blk0 > qmax_w) continue;                          // (wave-uniform) wholly above this wave's diagonal
        const char* sK = ring + (t & 3) * 8192;
        const char* sV = sK + 4096;
        f32x16 sa, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + swz(r, ks * 2 + h));
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + swz(r, ks * 2 + h));
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], dp, 0, 0, 0);
        }
        const bool edge = (g.causal && blk0 + 31 > qmin_w) || blk0 + 32 > Nk;   // (wave-uniform)
        const int lim = (g.causal ? min(s_q + 1, Nk) : Nk) - blk0 - 4 * h;
        const uint32_t kp0 = (uint32_t)(blk0 + 4 * h) >> 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float pr = __builtin_amdgcn_exp2f(fmaf(sa[i], g.c2, -lse2));
            if (edge && (i & 3) + 8 * (i >> 2) >= lim) pr = 0.f;
            float dpv = dp[i] * ks_scale;                                  // dP * keep / (1 - p) * scale
            if (g.thr) {
                const uint32_t x = drop_mix(g.dbase, (uint32_t)s_q, kp0 + (uint32_t)(((i & 3) + 8 * (i >> 2)) >> 1));
                if (((i & 1) ? (x >> 16) : (x & 0xffffu)) < g.thr) dpv = 0.f;
            }
            sa[i] = pr * (dpv - dels);                                     // dS[key][q] = P * (dP' - delta) * scale
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 lf = pack_acc(sa, s2);
#pragma unroll
            for (int dblk = 0; dblk < (D + 31) / 32; ++dblk) {
                const bf16x8 kt = col_frag(sK, 16 * s2 + 4 * h, dblk * 32, lane);   // K^T[d][key]
                dq[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, lf, dq[dblk], 0, 0, 0);
            }
        }
    }
    // dQ^T[d][q] -> rows through the wave's LDS slab; the ring is free now
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* slab = ring + w * 4096;
#pragma unroll
    for (int dblk = 0; dblk < (D + 31) / 32; ++dblk)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(dq[dblk][4 * gq + 0]) | ((uint32_t)f32_to_bf16(dq[dblk][4 * gq + 1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(dq[dblk][4 * gq + 2]) | ((uint32_t)f32_to_bf16(dq[dblk][4 * gq + 3]) << 16);
            *reinterpret_cast<uint2*>(slab + r * 128 + (((dblk * 8 + 2 * gq + h) ^ (r & 15)) << 3)) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* dqo = static_cast<bf16_t*>(a.dq) + g.b * a.bsdq + g.hh * D;
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
        const int row = p4 * 8 + (lane >> 3), q8 = lane & 7;
        uint4 raw = *reinterpret_cast<const uint4*>(slab + row * 128 + ((q8 ^ ((row & 15) >> 1)) << 4));
        if (row & 1) { const uint32_t a0 = raw.x, a1 = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = a0; raw.w = a1; }
        const int s_row = qmin_w + row;
        if (s_row < Nq && q8 < CH) *reinterpret_cast<uint4*>(dqo + (int64_t)s_row * a.lddq + q8 * 8) = raw;
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV.  Stream step = one chunk of 32 queries: ring stage = [Q rows 4 KiB | dO rows 4 KiB | lse[32] delta[32]];
// 9 DMA instructions per step (4 + 4 of 1 KiB, one of 256 B), dealt round-robin to the 4 waves.
// ------------------------------------------------------------------------------------------------
constexpr int QC = 32;

template <int D>
__global__ __launch_bounds__(256, 2) void flash_dkv_kernel(const focus_flash_args a) {
    constexpr int KS = D / 16, CH = D / 8, NW = 4;
    __shared__ __attribute__((aligned(1024))) char ring[4 * 8192 + 4 * 256];   // 4 x (Q | dO), then 4 x (lse | delta)

    int bx, bh;
    focus_xcd_group(bx, bh);
    const Geo g = geo_of<D>(a, bh);
    const int Nq = g.Nq, Nk = g.Nk;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int kw = bx * QT + w * 32;                       // first key of this wave's block
    const int key = kw + r;
    const bool key_ok = key < Nk;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Pack8 ka, va;
        ka.u = make_uint4(0, 0, 0, 0); va.u = ka.u;
        if (key_ok) {
            ka.u = *reinterpret_cast<const uint4*>(g.k + (int64_t)key * g.ldk + ks * 16 + 8 * h);
            va.u = *reinterpret_cast<const uint4*>(g.v + (int64_t)key * g.ldv + ks * 16 + 8 * h);
        }
        kf[ks] = ka.v; vf[ks] = va.v;
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ordinary loads done before any DMA is counted
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(kf[ks])); asm volatile("" : "+v"(vf[ks])); }

    const int nchunk = (Nq + QC - 1) / QC;
    const int ch0 = g.causal ? min(bx * QT / QC, nchunk) : 0;          // chunks wholly above the key tile see none of it
    const bf16_t* dob = static_cast<const bf16_t*>(a.dout) + g.b * a.bsdo + g.hh * D;
    const int64_t stat0 = ((int64_t)g.b * g.heads + g.hh) * Nq;
    const int drow = lane >> 3, dkey = ((drow & 2) << 1) | ((drow >> 1) & 2);
    const uint32_t ring_a = lds_addr_of(ring);
    constexpr int MINE = (9 + NW - 1) / NW;                // upper bound of DMA instructions per wave per step (3)
    constexpr int MINE_MIN = 9 / NW;                       // every wave issues at least this many (2); wave 0 one more
    auto dma_chunk = [&](int ch) __attribute__((always_inline)) {
        const uint32_t st = ring_a + (ch & 3) * 8192;
#pragma unroll
        for (int j = 0; j < MINE; ++j) {
            const int i = w + j * NW;                      // wave-uniform instruction index 0..8
            if (i < 8) {
                const int gq = i & 3;                      // 8-row group of the Q (i < 4) or dO (i >= 4) tile
                const int64_t s = min(ch * QC + gq * 8 + drow, Nq - 1);
                const int chunk = min((lane & 7) ^ (dkey | (gq & 1)), CH - 1);
                const bf16_t* src = i < 4 ? g.q + s * g.ldq + chunk * 8 : dob + s * a.lddo + chunk * 8;
                glds16(src, __builtin_amdgcn_readfirstlane(st + i * 1024));
            } else if (i == 8) {
                // lanes 0-31: lse of the chunk's queries, lanes 32-63: delta (4 B per lane, 256-B piece)
                const int64_t s = min(ch * QC + r, Nq - 1);
                const float* src = (h ? a.delta : a.lse) + stat0 + s;
                glds4(src, __builtin_amdgcn_readfirstlane(ring_a + 4 * 8192 + (ch & 3) * 256));
            }
        }
    };
    if (ch0 < nchunk) dma_chunk(ch0);
    if (ch0 + 1 < nchunk) dma_chunk(ch0 + 1);
    if (ch0 + 2 < nchunk) dma_chunk(ch0 + 2);
    const float ks_scale = g.keep_scale * g.scale;
    const uint32_t kpair = (uint32_t)key >> 1;
    const bool hi = key & 1;

    for (int ch = ch0; ch < nchunk; ++ch) {
        if (ch + 2 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MINE_MIN) : "memory");
        else if (ch + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINE_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // chunk ch visible to all; everyone is done with stage (ch-1) & 3
        if (ch + 3 < nchunk) dma_chunk(ch + 3);
        const int qc0 = ch * QC;
        if (g.causal && qc0 + QC - 1 < kw) continue;                      // (wave-uniform) every query precedes this wave's keys
        const char* tq = ring + (ch & 3) * 8192;
        const char* td = tq + 4096;
        const float* sl = reinterpret_cast<const float*>(ring + 4 * 8192 + (ch & 3) * 256);
        // S'[q][key] and dP'[q][key]: queries on the accumulator rows, this wave's keys on the lanes
        f32x16 sa, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(tq + swz(r, ks * 2 + h));
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(td + swz(r, ks * 2 + h));
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        // rows of the tile: q = qc0 + (i&3) + 8*(i>>2) + 4*h.  Rows past Nq (copies of the last row) and, under the causal
        // mask, rows before this lane's key must not reach dK / dV; keys past Nk only reach rows that are never stored.
        const bool edge = (g.causal && qc0 < kw + 32) || qc0 + QC > Nq;   // (wave-uniform)
        const int qlo = g.causal ? key - qc0 - 4 * h : -1000000, qhi = Nq - qc0 - 4 * h;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 l4 = *reinterpret_cast<const float4*>(sl + 8 * gq + 4 * h);
            const float4 d4 = *reinterpret_cast<const float4*>(sl + 32 + 8 * gq + 4 * h);
            const float ls[4] = {l4.x * LOG2E, l4.y * LOG2E, l4.z * LOG2E, l4.w * LOG2E}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * gq + e, rowq = e + 8 * gq;              // = (i&3) + 8*(i>>2)
                float pr = __builtin_amdgcn_exp2f(fmaf(sa[i], g.c2, -ls[e]));
                if (edge && (rowq < qlo || rowq >= qhi)) pr = 0.f;
                float keepf = g.keep_scale;
                if (g.thr) {
                    const uint32_t x = drop_mix(g.dbase, (uint32_t)(qc0 + rowq + 4 * h), kpair);
                    if ((hi ? (x >> 16) : (x & 0xffffu)) < g.thr) keepf = 0.f;
                }
                sa[i] = pr * keepf;                                        // Pd[q][key]
                dp[i] = pr * fmaf(dp[i], keepf * g.scale, -de[e]);         // dS[q][key]
            }
        }
        (void)ks_scale;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack_acc(sa, s2), lf = pack_acc(dp, s2);
#pragma unroll
            for (int dblk = 0; dblk < (D + 31) / 32; ++dblk) {
                const bf16x8 dxc = col_frag(td, 16 * s2 + 4 * h, dblk * 32, lane);   // dO[q][d] by columns
                const bf16x8 qc = col_frag(tq, 16 * s2 + 4 * h, dblk * 32, lane);    // Q[q][d] by columns
                dv[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, dxc, dv[dblk], 0, 0, 0);
                dk[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf, qc, dk[dblk], 0, 0, 0);
            }
        }
    }
    // dK[key][d], dV[key][d]: accumulator row = key (in-block), column (lane) = d
    bf16_t* dko = static_cast<bf16_t*>(a.dk) + g.b * a.bsdk + g.hh * D;
    bf16_t* dvo = static_cast<bf16_t*>(a.dv) + g.b * a.bsdv + g.hh * D;
#pragma unroll
    for (int dblk = 0; dblk < (D + 31) / 32; ++dblk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kl = kw + (i & 3) + 8 * (i >> 2) + 4 * h, d = dblk * 32 + r;
            if (kl < Nk && d < D) {
                dko[(int64_t)kl * a.lddk + d] = f32_to_bf16(dk[dblk][i]);
                dvo[(int64_t)kl * a.lddv + d] = f32_to_bf16(dv[dblk][i]);
            }
        }
}

bool args_ok(const focus_flash_args* a, bool bwd) {
    if (!a || !a->q || !a->k || !a->v || !a->out || !a->lse) return false;
    if (a->drop_thr && !a->seed) return false;
    if (bwd && (!a->dout || !a->delta || !a->dq || !a->dk || !a->dv)) return false;
    return true;
}

bool strides_ok(const focus_flash_args* a, bool bwd) {
    const int64_t s[] = {a->ldq, a->ldk, a->ldv, a->ldo, a->bsq, a->bsk, a->bsv, a->bso};
    for (int64_t v : s)
        if (v & 7) return false;
    if (!focus_aligned(a->q, 16) || !focus_aligned(a->k, 16) || !focus_aligned(a->v, 16) || !focus_aligned(a->out, 16)) return false;
    if (bwd) {
        const int64_t t[] = {a->lddo, a->lddq, a->bsdo, a->bsdq};
        for (int64_t v : t)
            if (v & 7) return false;
        if (!focus_aligned(a->dout, 16) || !focus_aligned(a->dq, 16)) return false;
    }
    return true;
}

}  // namespace

int focus_flash_attn_ok(int Nq, int Nk, int d, int dtype, int causal) {
    static const bool enabled = !(getenv("FOCUS_FLASH") && atoi(getenv("FOCUS_FLASH")) == 0);
    if (!enabled || dtype != FOCUS_BF16 || (d != 32 && d != 48 && d != 64) || Nq < 1 || Nk < 1) return 0;
    if (causal && Nq != Nk) return 0;
    return 1;
}

int focus_flash_attn_fwd(const focus_flash_args* a, void* stream) {
    if (!args_ok(a, false)) return FOCUS_ERR_NULL;
    if (!focus_flash_attn_ok(a->Nq, a->Nk, a->d, a->dtype, a->causal) || a->B < 1 || a->heads < 1 || a->drop_thr >= 65536u ||
        (int64_t)a->B * a->heads > 65535)
        return FOCUS_ERR_SHAPE;
    if (!strides_ok(a, false)) return FOCUS_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((a->Nq + QT - 1) / QT, a->B * a->heads);
    // 128-key tiles when the sequence has them (fewer barriers per product), 64 otherwise
#define FWD(D) do { if (a->Nk > 256) hipLaunchKernelGGL((flash_fwd_kernel<D, 4>), grid, dim3(256), 0, s, *a); \
                    else hipLaunchKernelGGL((flash_fwd_kernel<D, 2>), grid, dim3(256), 0, s, *a); } while (0)
    if (a->d == 32) FWD(32); else if (a->d == 48) FWD(48); else FWD(64);
#undef FWD
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

int focus_flash_attn_bwd(const focus_flash_args* a, void* stream) {
    if (!args_ok(a, true)) return FOCUS_ERR_NULL;
    if (!focus_flash_attn_ok(a->Nq, a->Nk, a->d, a->dtype, a->causal) || a->B < 1 || a->heads < 1 || a->drop_thr >= 65536u ||
        (int64_t)a->B * a->heads > 65535)
        return FOCUS_ERR_SHAPE;
    if (!strides_ok(a, true)) return FOCUS_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    dim3 gq((a->Nq + QT - 1) / QT, a->B * a->heads), gk((a->Nk + QT - 1) / QT, a->B * a->heads);
#define BWD(D) do { hipLaunchKernelGGL((flash_dq_kernel<D>), gq, dim3(256), 0, s, *a); \
                    hipLaunchKernelGGL((flash_dkv_kernel<D>), gk, dim3(256), 0, s, *a); } while (0)
    if (a->d == 32) BWD(32); else if (a->d == 48) BWD(48); else BWD(64);
#undef BWD
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
