// traj_space_bwd_mfma.hip -- fused backward of the space step of trajectory attention (bf16, head dim 64).
//
// Flash-style: nothing of size S x S is stored.  With P[s,f,p] = exp(scale*q_s.k_{f,p} - lse[s,f]) recomputed from
// the forward's log-sum-exp, dX = dx~ (+ dx_diag on the query's own frame), delta[s,f] = dX[s,f,:].x~[s,f,:]:
//     dV_{f,p} = sum_s P * dX[s,f,:]          dP = dX[s,f,:].v_{f,p}       dL = P * (dP - delta) * scale
//     dK_{f,p} = sum_s dL * q_s               dQ_s = sum_{f,p} dL * k_{f,p}
// Three kernels, no atomics, every output written exactly once:
//   traj_dxsum_kernel : dxsum [B,S,C] = the dX rows of each query's own frame (dx~ + dx_diag): 3 x 19 MB
//   traj_dq_kernel    : [r3] also forms delta: it streams its queries' dX rows anyway, the matching x~ rows come beside
//                       them (one more 4 KiB tile per wave and frame) and delta = rowsum(dX . x~) is 32 FMAs and one
//                       cross-half add per lane and frame; written to [B,h,S,F] for the dK/dV kernel.  The separate
//                       traj_delta_kernel (a full pass over dx~ AND x~: 408 MB of HBM traffic per launch, 76 us) is gone.
//                       workgroup = 128 queries of one (b,h), walks the (frame, 32-key block) stream; keys on the
//                       accumulator rows (swapped products), dL feeds  dQ^T += K^T.dL  straight from the accumulator
//                       registers; K^T fragments come from the row-major K block through ds_read_b64_tr_b16.
//   traj_dkv_kernel   : workgroup = the <=224 keys of one (b,h,frame), one wave per 32 keys, walks all queries in
//                       32-query chunks; queries on the accumulator rows, P and dL feed  dV += P^T.dX,  dK += dL^T.Q
//                       as A operands, dX / Q column fragments through ds_read_b64_tr_b16.
// Both MFMA kernels stream their tiles through a 4-stage LDS ring filled by LDS-DMA three steps ahead (inline-asm
// global_load_lds_dwordx4, counted s_waitcnt vmcnt, one raw s_barrier per step): no vector-memory load sits in the
// loops, so nothing but the DMA itself is on the vmcnt counter and the compiler adds no waits of its own.
#include "focus_common.h"
#include "traj_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int HD = 64;
constexpr int QT = 128;
constexpr float LOG2E = 1.44269504088896341f;

union Pack8 { bf16x8 v; bf16_t e[8]; uint4 u; uint2 h2[2]; s16x4 t[2]; };

// 16-B chunk swizzle of a [rows][64 bf16] tile with 128-B rows.  A row covers half of the 64 LDS banks (row parity
// picks the half), so the XOR key is built from row>>1, bit-reversed: (a) a ds_read_b128 bank group reads one chunk
// of 16 rows whose row>>1 values are 8 distinct ones mod 8 -> 16 distinct (half, chunk) slots; (b) a
// ds_read_b64_tr_b16 bank group reads 4 adjacent chunks of rows r..r+3 -> rows r and r+2 need keys that differ in
// bit 2, which the reversal provides.  (key = row & 7 was 2-way conflicted on both: SQ_LDS_BANK_CONFLICT ~40 %.)
__device__ __forceinline__ int swz(int row, int chunk) {
    const int key = ((row & 2) << 1) | ((row >> 1) & 2) | ((row >> 3) & 1);
    return row * 128 + ((chunk ^ key) << 4);
}

// MFMA 32x32x16 operand whose k index runs over tile ROWS and whose m/n index is a tile COLUMN, in the k order of
// an accumulator tile used as the other operand: element j <-> row r0 + 8*(j>>2) + (j&3), column c0 + (lane&31),
// where the caller passes r0 = 16*s + 4*(lane>>5) (+ block base).  Two transposed LDS reads (T10).
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

// ------------------------------------------------------------------------------------------------
// dxsum[b,s,:] = dxt[b,s,s/P,:] + dxdiag[b,s,:]      (so the kernels below can DMA dX rows without an add)
// one thread per 16-byte chunk of a [B*S, C] row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void traj_dxsum_kernel(const bf16_t* __restrict__ dxt, const bf16_t* __restrict__ dxdiag,
                                                         bf16_t* __restrict__ dxsum, int64_t nchunks, int S, int F, int P, int C) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nchunks) return;
    const int cpr = C / 8;
    const int64_t row = g / cpr;
    const int ch = (int)(g - row * cpr);
    const int s = (int)(row % S), fs = s / P;
    Pack8 a, d2, o;
    a.u = *reinterpret_cast<const uint4*>(dxt + ((row * F + fs) * cpr + ch) * 8);
    d2.u = *reinterpret_cast<const uint4*>(dxdiag + g * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o.e[j] = f32_to_bf16(bf16_to_f32(a.e[j]) + bf16_to_f32(d2.e[j]));
    *reinterpret_cast<uint4*>(dxsum + g * 8) = o.u;
}

constexpr int MAXF = 16;     // frames the per-wave lse / delta tables are sized for

// ------------------------------------------------------------------------------------------------
// dQ.  Stream step t = (frame f, key block kb): ring stage t & 3 = [K block 32 x 128 B | V block 32 x 128 B]; wave w
// DMAs rows 8w..8w+7 of both (2 instructions per step).  The wave's own 32 dX rows of frame f come by DMA into a
// private 4 KiB tile (4 instructions per frame, issued at kb = 0 of frame f-1 after the fragments of frame f-1 have
// been read into registers); lse / delta of its queries for all frames sit in LDS tables filled before the loop.
// Wait counts: an operation has landed once at most (instructions issued after it) remain outstanding.
// ------------------------------------------------------------------------------------------------
template <int NKB>
__global__ __launch_bounds__(256, 2) void traj_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dxt,
                                                         const bf16_t* __restrict__ dxsum, const bf16_t* __restrict__ xt,
                                                         const float* __restrict__ lse, float* __restrict__ delta,
                                                         bf16_t* __restrict__ dqkv, int B, int F, int P, int heads) {
    __shared__ __attribute__((aligned(1024))) char ring[4 * 8192];       // after the loop: the 4 output slabs
    __shared__ __attribute__((aligned(1024))) char sDX[4 * 4096];        // per wave: [32 q][128 B] dX rows, swizzled
    __shared__ __attribute__((aligned(1024))) char sXT[4 * 4096];        // per wave: the same rows of x~ (for delta)
    __shared__ float sLse[4][MAXF * 32];

    const int S = F * P, N = S + 1, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    int bx, bh;
    focus_xcd_group(bx, bh);                                   // the query tiles of one (b, h) share an XCD's L2
    const int b = bh / heads, hh = bh % heads;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int s_raw = bx * QT + w * 32 + r;
    const bool q_valid = s_raw < S;
    const int s_q = min(s_raw, S - 1);
    const bf16_t* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;
    const int T = F * NKB;

    // ---- DMA helpers ----
    const int drow = lane >> 3, dkey = ((drow & 2) << 1) | ((drow >> 1) & 2), dchunk = lane & 7;
    const uint32_t ring_a = lds_addr_of(ring), dx_a = lds_addr_of(sDX) + w * 4096, xt_a = lds_addr_of(sXT) + w * 4096;
    auto dma_step = [&](int t) __attribute__((always_inline)) {          // K and V rows 8w..8w+7 of block (f, kb)
        const int f = t / NKB, kb = t - f * NKB;
        const int row = min(kb * 32 + w * 8 + drow, P - 1);              // padded keys: a copy of the last real row
        const bf16_t* src = base + (int64_t)(1 + f * P + row) * tok + C + ((dchunk ^ (dkey | (w & 1))) << 3);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_a + (t & 3) * 8192 + w * 1024);
        glds16(src, dst);
        glds16(src + C, dst + 4096);
    };
    auto dma_dx = [&](int f) __attribute__((always_inline)) {            // this wave's 32 dX rows and x~ rows of frame f
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s = min(bx * QT + w * 32 + g * 8 + drow, S - 1);
            const bool own = s >= f * P && s < (f + 1) * P;
            const int64_t rf = (((int64_t)b * S + s) * F + f) * C;
            const bf16_t* rowp = own ? dxsum + ((int64_t)b * S + s) * C : dxt + rf;
            const int cofs = hh * HD + ((dchunk ^ (dkey | (g & 1))) << 3);
            glds16(rowp + cofs, __builtin_amdgcn_readfirstlane(dx_a + g * 1024));
            glds16(xt + rf + cofs, __builtin_amdgcn_readfirstlane(xt_a + g * 1024));
        }
    };

    // ---- prologue: tables, Q fragments, first DMAs ----
    const int64_t sf0 = (((int64_t)b * heads + hh) * S + s_q) * F;
    for (int f2 = h; f2 < F; f2 += 2)
        sLse[w][f2 * 32 + r] = q_valid ? lse[sf0 + f2] * LOG2E : INFINITY;   // base-2 units; +inf -> P = 0 for padded queries
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)(1 + s_q) * tok + ks * 16 + 8 * h);
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ordinary loads above are done before any DMA is counted
    // ... and the compiler is told so: without this it keeps the Q loads "pending" into the loop and emits its own
    // s_waitcnt vmcnt(0) at their first use in EVERY step, draining the DMA ring
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));
    dma_dx(0);
    dma_step(0);
    if (T > 1) dma_step(1);
    if (T > 2) dma_step(2);

    const char* mydx = sDX + w * 4096;
    const char* myxt = sXT + w * 4096;
    bf16x8 df[4];
    float lse2 = 0.f, dels = 0.f;
    int t = 0;
    for (int f = 0; f < F; ++f) {
        const bool more = f + 1 < F;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb, ++t) {
            // ---- step t landed?  instructions issued after it: steps t+1, t+2 (2 each) and, for kb = 1..3, the
            // 8 dX / x~ instructions of frame f+1 issued at kb = 0 ----
            // (spelled out as compile-time counts: a run-time switch here cost a maze of ~100 scalar instructions per step)
            if (t + 2 < T) {
                if (kb == 0) { if (NKB >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                else if (kb <= 3 && more) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if (t + 1 < T && NKB >= 4) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // step t visible to all; everyone is done with stage (t-1) & 3
            if (kb == 0) {
                // this frame's dX fragments (B operand: [k=d][col=q]) and softmax statistics
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) df[ks] = *reinterpret_cast<const bf16x8*>(mydx + swz(r, ks * 2 + h));
                lse2 = sLse[w][f * 32 + r];
                // delta[q, f] = scale * sum_d dX[q,d] x~[q,d]: this lane holds 32 of the 64 channels of query r
                float dsum = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    Pack8 xa, da;
                    xa.v = *reinterpret_cast<const bf16x8*>(myxt + swz(r, ks * 2 + h));
                    da.v = df[ks];
#pragma unroll
                    for (int j = 0; j < 8; ++j) dsum = fmaf(bf16_to_f32(da.e[j]), bf16_to_f32(xa.e[j]), dsum);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the dX / x~ tiles are free for frame f+1
                dsum += __shfl_xor(dsum, 32, 64);
                dels = dsum * scale;
                if (h == 0 && q_valid) delta[sf0 + f] = dels;            // for the dK / dV kernel (already times scale)
            }
            if (t + 3 < T) dma_step(t + 3);
            if (kb == 0 && more) dma_dx(f + 1);

            const char* sK = ring + (t & 3) * 8192;
            const char* sV = sK + 4096;
            f32x16 sa, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + swz(r, ks * 2 + h));
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + swz(r, ks * 2 + h));
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float pr = __builtin_amdgcn_exp2f(fmaf(sa[i], c2, -lse2));
                if (kb == NKB - 1) {               // padded keys hold a copy of the last real key: drop them
                    if ((i & 3) + 8 * (i >> 2) >= P - kb * 32 - 4 * h) pr = 0.f;
                }
                sa[i] = pr * fmaf(dp[i], scale, -dels);   // dL[key][q] = P * (dP - delta) * scale
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 lf = pack_acc(sa, s2);
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) {
                    const bf16x8 kt = col_frag(sK, 16 * s2 + 4 * h, dblk * 32, lane);   // K^T[d][key]
                    dq[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, lf, dq[dblk], 0, 0, 0);
                }
            }
        }
    }
    // dQ^T[d][q] -> rows of dqkv through the wave's LDS slab (same scheme as the forward); the ring is free now
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* slab = ring + w * 4096;
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(dq[dblk][4 * g + 0]) | ((uint32_t)f32_to_bf16(dq[dblk][4 * g + 1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(dq[dblk][4 * g + 2]) | ((uint32_t)f32_to_bf16(dq[dblk][4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(slab + r * 128 + (((dblk * 8 + 2 * g + h) ^ (r & 15)) << 3)) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
        const int row = p4 * 8 + (lane >> 3), q8 = lane & 7;
        uint4 raw = *reinterpret_cast<const uint4*>(slab + row * 128 + ((q8 ^ ((row & 15) >> 1)) << 4));
        if (row & 1) { const uint32_t a0 = raw.x, a1 = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = a0; raw.w = a1; }
        const int s_row = bx * QT + w * 32 + row;
        if (s_row < S)
            *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + 1 + s_row) * tok + hh * HD + q8 * 8) = raw;
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV.  Stream step = one chunk of 32 queries: ring stage = [Q rows 4 KiB | dX rows 4 KiB | lse[32] del[32]];
// 9 DMA instructions per step (4 + 4 of 1 KiB, one of 256 B), dealt round-robin to the NKB waves.
// ------------------------------------------------------------------------------------------------
constexpr int QC = 32;   // queries per chunk

template <int NKB>
__global__ __launch_bounds__(64 * NKB) void traj_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dxt,
                                                            const bf16_t* __restrict__ dxsum,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int B, int F, int P, int heads,
                                                            int NT) {
    __shared__ __attribute__((aligned(1024))) char ring[4 * 8192 + 4 * 256];   // 4 x (Q | dX), then 4 x (lse | del)

    const int S = F * P, N = S + 1, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    // workgroup = key tile jt (NKB blocks of 32 keys) of frame f; a frame has NT tiles (1 unless P > 224)
    int bx, bh;
    focus_xcd_group(bx, bh);                                   // the (frame, key tile) workgroups of one (b, h) share an L2
    const int f = bx / NT, jt = bx - f * NT, b = bh / heads, hh = bh % heads;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;

    // this wave's 32 keys as B operands [k=d][col=key]: kept in registers for the whole sweep
    const int kw = (jt * NKB + w) * 32;                   // first key (in the frame) of this wave's block
    const int key = kw + r;
    const bool key_ok = key < P;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        Pack8 a, c;
        a.u = make_uint4(0, 0, 0, 0); c.u = a.u;
        if (key_ok) {
            const bf16_t* row = base + (int64_t)(1 + f * P + key) * tok + ks * 16 + 8 * h;
            a.u = *reinterpret_cast<const uint4*>(row + C);
            c.u = *reinterpret_cast<const uint4*>(row + 2 * C);
        }
        kf[ks] = a.v; vf[ks] = c.v;
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ordinary loads done before any DMA is counted
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+v"(kf[ks])); asm volatile("" : "+v"(vf[ks])); }   // (see dq)

    const int nchunk = (S + QC - 1) / QC;
    const int drow = lane >> 3, dkey = ((drow & 2) << 1) | ((drow >> 1) & 2), dchunk = lane & 7;
    const uint32_t ring_a = lds_addr_of(ring);
    constexpr int MINE = (9 + NKB - 1) / NKB;              // upper bound of DMA instructions per wave per step
    constexpr int MINE_MIN = 9 / NKB > 4 ? 4 : 9 / NKB;    // every wave issues at least this many (waves w < 9 % NKB one more); capped: vmcnt(N) <= 8
    auto dma_chunk = [&](int ch) __attribute__((always_inline)) {
        const uint32_t st = ring_a + (ch & 3) * 8192;
#pragma unroll
        for (int j = 0; j < MINE; ++j) {
            const int i = w + j * NKB;                     // wave-uniform instruction index 0..8
            if (i < 8) {
                const int g = i & 3;                       // 8-row group of the Q (i < 4) or dX (i >= 4) tile
                const int s = min(ch * QC + g * 8 + drow, S - 1);
                const int cofs = (dchunk ^ (dkey | (g & 1))) << 3;
                const bf16_t* src;
                if (i < 4) src = base + (int64_t)(1 + s) * tok + cofs;
                else src = ((s >= f * P && s < (f + 1) * P) ? dxsum + ((int64_t)b * S + s) * C
                                                            : dxt + (((int64_t)b * S + s) * F + f) * C) + hh * HD + cofs;
                glds16(src, __builtin_amdgcn_readfirstlane(st + i * 1024));
            } else if (i == 8) {
                // lanes 0-31: lse of the chunk's queries, lanes 32-63: delta (4 B per lane, 256-B piece)
                const int s = min(ch * QC + r, S - 1);
                const float* src = (h ? delta : lse) + (((int64_t)b * heads + hh) * S + s) * F + f;
                glds4(src, __builtin_amdgcn_readfirstlane(ring_a + 4 * 8192 + (ch & 3) * 256));
            }
        }
    };
    dma_chunk(0);
    if (nchunk > 1) dma_chunk(1);
    if (nchunk > 2) dma_chunk(2);

    for (int ch = 0; ch < nchunk; ++ch) {
        // every wave issued at least MINE_MIN instructions per chunk: waiting down to that many per chunk still in
        // flight is exact for the waves that issued MINE_MIN and slightly early-safe for the others
        if (ch + 2 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MINE_MIN) : "memory");
        else if (ch + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINE_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // chunk ch visible to all; everyone is done with stage (ch-1) & 3
        if (ch + 3 < nchunk) dma_chunk(ch + 3);
        const char* tq = ring + (ch & 3) * 8192;
        const char* td = tq + 4096;
        const float* sl = reinterpret_cast<const float*>(ring + 4 * 8192 + (ch & 3) * 256);
        // S'[q][key] and dP'[q][key]: queries on the accumulator rows, this wave's keys on the lanes
        f32x16 sa, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(tq + swz(r, ks * 2 + h));
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(td + swz(r, ks * 2 + h));
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        // rows of the tile: q = (i&3) + 8*(i>>2) + 4*h  -> 4 consecutive floats per group of 4 registers.
        // lse arrives in natural-log units (the forward's), delta times scale (traj_dq_kernel).  Keys past P need no mask:
        // they only reach accumulator rows that are never stored.
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(sl + 8 * g + 4 * h);
            const float4 d4 = *reinterpret_cast<const float4*>(sl + 32 + 8 * g + 4 * h);
            const float ls[4] = {l4.x * LOG2E, l4.y * LOG2E, l4.z * LOG2E, l4.w * LOG2E}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                const float pr = __builtin_amdgcn_exp2f(fmaf(sa[i], c2, -ls[e]));
                sa[i] = pr;                                        // P[q][key]
                dp[i] = pr * fmaf(dp[i], scale, -de[e]);           // dL[q][key]
            }
        }
        if (ch == nchunk - 1 && (S & (QC - 1))) {
            // queries past S (rows that are copies of row S-1) must not reach dK / dV
            const int qlim = S - ch * QC - 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & 3) + 8 * (i >> 2) >= qlim) { sa[i] = 0.f; dp[i] = 0.f; }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack_acc(sa, s2), lf = pack_acc(dp, s2);
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk) {
                const bf16x8 dxc = col_frag(td, 16 * s2 + 4 * h, dblk * 32, lane);   // dX[q][d] by columns
                const bf16x8 qc = col_frag(tq, 16 * s2 + 4 * h, dblk * 32, lane);    // Q[q][d] by columns
                dv[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, dxc, dv[dblk], 0, 0, 0);
                dk[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf, qc, dk[dblk], 0, 0, 0);
            }
        }
    }
    // dK[key][d], dV[key][d]: accumulator row = key (in-block), column (lane) = d
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kl = kw + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (kl < P) {
                bf16_t* row = dqkv + ((int64_t)b * N + 1 + f * P + kl) * tok + hh * HD + dblk * 32 + r;
                row[C] = f32_to_bf16(dk[dblk][i]);
                row[2 * C] = f32_to_bf16(dv[dblk][i]);
            }
        }
}

template <int NB>
int launch_dq(const void* qkv, const void* dxt, const void* dxsum, const void* xt, const float* lse, float* delta, void* dqkv,
              int B, int F, int P, int heads, hipStream_t s) {
    const int S = F * P;
    hipLaunchKernelGGL((traj_dq_kernel<NB>), dim3((S + QT - 1) / QT, B * heads), dim3(256), 0, s, (const bf16_t*)qkv,
                       (const bf16_t*)dxt, (const bf16_t*)dxsum, (const bf16_t*)xt, lse, delta, (bf16_t*)dqkv, B, F, P, heads);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <int NKB>
int launch_dkv(const void* qkv, const void* dxt, const void* dxsum, const float* lse, const float* delta, void* dqkv,
               int B, int F, int P, int heads, int NT, hipStream_t s) {
    hipLaunchKernelGGL((traj_dkv_kernel<NKB>), dim3(F * NT, B * heads), dim3(64 * NKB), 0, s, (const bf16_t*)qkv,
                       (const bf16_t*)dxt, (const bf16_t*)dxsum, lse, delta, (bf16_t*)dqkv, B, F, P, heads, NT);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

// Patch-token rows of dqkv (q, k and v parts of tokens 1..N-1) are fully written; the cls row/parts are the caller's.
// delta: [B,h,S,F] fp32 scratch (written by the dQ kernel, read by the dK/dV kernel); dxsum: [B,S,C] bf16 scratch;
// lse2 is unused since round 3 (kept in the signature: the workspace layout of focus_traj_space_bwd is unchanged).
int focus_traj_space_bwd_mfma(const void* qkv, const void* xt, const float* lse, const void* dxt, const void* dxdiag,
                              float* delta, float* lse2, void* dxsum, void* dqkv, int B, int F, int P, int heads,
                              hipStream_t s) {
    (void)lse2;
    if (B * heads > 65535 || F > MAXF || P > 32 * FOCUS_TRAJ_MAX_KEY_BLOCKS) return FOCUS_ERR_SHAPE;
    const int S = F * P, C = heads * HD;
    const int64_t nchunks = (int64_t)B * S * (C / 8);
    hipLaunchKernelGGL(traj_dxsum_kernel, dim3((unsigned)cdiv64(nchunks, 256)), dim3(256), 0, s, (const bf16_t*)dxt,
                       (const bf16_t*)dxdiag, (bf16_t*)dxsum, nchunks, S, F, P, C);
    FOCUS_CHECK_LAUNCH();
    int rc = FOCUS_ERR_SHAPE;
    // dQ streams the frame's ceil(P/32) key blocks (exact count: the tail mask sits in the last one) and writes delta
#define DQ(K) case K: rc = launch_dq<K>(qkv, dxt, dxsum, xt, lse, delta, dqkv, B, F, P, heads, s); break
    switch ((P + 31) / 32) {
        DQ(1); DQ(2); DQ(3); DQ(4); DQ(5); DQ(6); DQ(7); DQ(8); DQ(9); DQ(10); DQ(11); DQ(12); DQ(13); DQ(14);
        default: break;
    }
#undef DQ
    if (rc) return rc;
    // dK/dV: one workgroup per (frame, key tile of NKB blocks), one wave per block
    int nkb, nt;
    focus_traj_space_tiling(P, &nkb, &nt);
#define DKV(K) case K: return launch_dkv<K>(qkv, dxt, dxsum, lse, delta, dqkv, B, F, P, heads, nt, s)
    switch (nkb) {
        DKV(1); DKV(2); DKV(3); DKV(4); DKV(5); DKV(6); DKV(7);
        default: break;
    }
#undef DKV
    return FOCUS_ERR_SHAPE;
}
