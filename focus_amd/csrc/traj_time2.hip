// traj_time2.hip -- temporal step of trajectory attention (attention.py:536-549) WITHOUT k2 = proj_kv(x~) in HBM.
//
// With use_original_code=True (the only mode the reference runs) the temporal logits are
//     scale * q2[s,h,:] . (Wk[h] x~[s,f,:] + bk[h])  =  scale * u[s,h,:] . x~[s,f,:]  +  (a term constant in f),
//     u[s,h,:] = Wk[h]^T q2[s,h,:]  in R^C,
// and the softmax over the F frames is shift invariant: k2 [B,S,F,C] (the block's largest GEMM, 8x the tokens, and the
// 154 MB tensor it writes and the time kernels re-read, plus dk2 of the same size backward) never has to exist, and
// proj_kv.bias gets its exact zero gradient.  u is 12x768 numbers per query -- larger than k2's row -- so it is never
// written either: it is formed per 64-channel chunk ON CHIP.
//
// Decomposition (the same for forward and backward): workgroup = (channel chunk cc of 64 channels, query range).
// Its slice Wk^T[cc*64 .. +63][all (h,dd)] (96 KB bf16) sits in LDS for the whole launch; per tile of 16 queries:
//     U^T[c, s]   = sum_dd WkT[c, h*64+dd] q2[s, h*64+dd]          MFMA 16x16x32, M = channels, N = queries, K = 64
//                   -> bf16 -> LDS image U[s][h][c]
//   forward  (time2_logits_kernel):
//     L[f, h]    += sum_c x~[s,f,c] U[s,h,c]   per query s            MFMA 16x16x32, M = frames, N = heads, K = 64;
//                   the A operand is read from x~ in HBM directly (16 B per lane, 128-B row segments); partial
//                   logits of the chunk go to slab[cc][row][h][f]; time2_softmax_kernel sums the C/64 slabs and takes
//                   the softmax over f (attn2 [B,S,h,F]); time2_out_kernel forms out = sum_f a x~ (reads x~ once more).
//   backward (time2_dl_kernel, then time2_dx_kernel):
//     dl[s,f,h]   = scale a (da - sum_f a da),  da[s,f,h] = dout[s,h,:] . x~[s,f,h,:]
//     dx~[s,f,c]  = a[s,f,h(c)] dout[s,c] + sum_h dl[s,f,h] U[s,h,c]                      (VALU, U from LDS)
//     g[s,h,c]    = sum_f dl[s,f,h] x~[s,f,c]   ( = du )  -> HBM [B,S,h,C] bf16: the host forms
//                   dq2 = scale^-1-free GEMM  g[:,h,:] . Wk[h]^T  and  dWk[h] = q2[:,h,:]^T . g[:,h,:]  from it.
// HBM per block and direction: x~ is read twice forward (logits, out) and twice backward (dl, g), dx~ and g are
// written once; the k2 path moved k2/dk2 four more times and ran three 100352 x 768 x 768 GEMMs.
#include "focus_common.h"
#include <cstdlib>
#include "traj_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int CH = 64;                 // channels per chunk (= head dim: chunk cc holds exactly head cc's channels)
constexpr int TQ = 16;                 // queries per tile (MFMA N)
constexpr int MAXH = 16;               // heads <= 16 (MFMA N of the logit product)
constexpr int MAXF = 16;               // frames <= 16 (MFMA M of the logit product)
constexpr int NGRP = 2;                // wave groups (4 waves each) per workgroup, each streaming its own tiles
constexpr int HPW = MAXH / 4;          // heads per wave (h = w, w + 4, ...)

union Pk4 { uint2 u; bf16_t e[4]; };

__device__ __forceinline__ void unpack8(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    return o;
}
__device__ __forceinline__ float sum8lanes(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also releases global memory at workgroup scope, for which
// hipcc drains s_waitcnt vmcnt(0): that would wait for the NEXT tile's prefetch loads (and this tile's stores) at every
// barrier and serialise the stream.  The tiles only hand LDS data between waves.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS images, padded against bank conflicts (modelled per instruction with the lane groups of MI355X_MICROARCH.md "LDS";
// SQ_LDS_BANK_CONFLICT was 57 % of the LDS cycles with +16-byte rows):
//   W rows + 32 B: a ds_read_b128 lane group mixes 8 rows of k-group a with 8 rows of k-group a + 1; a row pitch of
//                  2 slots (mod 16) keeps their 16-byte slots apart (pitch of 1 slot: one 2-way conflict per group).
//   U [16 s][heads][64 c]: query rows + 16 B (the MFMA result is written with the query on the lane); in the forward
//                  the head rows are read as the B operand with the HEAD on the lane: head pitch 128 + 16 B.
__host__ __device__ __forceinline__ int wrow_bytes(int C) { return C * 2 + 32; }              // Wk^T slice: [64 c][C (h,dd)]
template <int HPAD> __host__ __device__ __forceinline__ int urow_bytes(int heads) { return heads * (CH * 2 + HPAD) + 16; }

// Wk^T slice of this chunk -> LDS (once per workgroup; register staged: the padded rows rule out LDS-DMA)
__device__ __forceinline__ void load_w_slice(char* sW, const bf16_t* wkT, int64_t ldw, int cc, int C) {
    const int pieces = C / 8, wrow = wrow_bytes(C), total = CH * pieces;      // 16-byte pieces per row / in the slice
    constexpr int UN = 6;                                        // loads in flight per thread (a serial load -> ds_write
    for (int e0 = threadIdx.x; e0 < total; e0 += UN * blockDim.x) {   // loop costs one L2/HBM round trip per piece)
        uint4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = min(e0 + u * (int)blockDim.x, total - 1);
            const int row = e / pieces, pc = e - row * pieces;
            v[u] = *reinterpret_cast<const uint4*>(wkT + (int64_t)(cc * CH + row) * ldw + pc * 8);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * (int)blockDim.x;
            if (e < total) {
                const int row = e / pieces, pc = e - row * pieces;
                *reinterpret_cast<uint4*>(sW + row * wrow + pc * 16) = v[u];
            }
        }
    }
}

// q2 fragments (MFMA B operand: column = query, k = dd) of this wave's heads for the tile starting at row0
struct QFrag { bf16x8 v[HPW][2]; };
__device__ __forceinline__ void load_q_frags(QFrag& q, const bf16_t* __restrict__ q2, int row0, int rows, int C, int heads,
                                             int wg, int lane) {
    const int col = lane & 15, kg = lane >> 4;
    const int64_t qoff = (int64_t)min(row0 + col, rows - 1) * C;  // this lane's query (MFMA column), clamped
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        const int h = min(wg + 4 * i, heads - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            q.v[i][ks] = *reinterpret_cast<const bf16x8*>(q2 + qoff + h * CH + 32 * ks + 8 * kg);
    }
}

// U[s][h][c] of one 16-query tile for the heads of this wave (h = wg, wg + 4, ...): U^T = WkT_slice . q2^T per head.
template <int HPAD>
__device__ __forceinline__ void compute_u_tile(const char* sW, char* sU, const QFrag& q, int C, int heads, int wg, int lane) {
    const int wrow = wrow_bytes(C), urow = urow_bytes<HPAD>(heads);
    const int col = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        const int h = wg + 4 * i;
        if (h >= heads) break;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sW + (16 * mt + col) * wrow + (h * CH + 32 * ks + 8 * kg) * 2);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, q.v[i][ks], acc, 0, 0, 0);
            }
            // acc[r] = U^T[c = 16 mt + 4 kg + r][s = col]
            Pk4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p.e[r] = f32_to_bf16(acc[r]);
            *reinterpret_cast<uint2*>(sU + col * urow + h * (CH * 2 + HPAD) + (16 * mt + 4 * kg) * 2) = p.u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward 1/3: partial logits of one channel chunk.  512 threads = 2 groups of 4 waves; group g streams the tiles
// t_begin + g, t_begin + g + 2, ... of the workgroup's range with its own U buffer (twice the bytes in flight of one
// group: with the weight slice in LDS only one workgroup fits a CU, and ~16 KB of x~ per tile against 2-4 us of HBM
// latency is what bounds the kernel).  A wave's x~ fragments pack 16 / F queries into the 16 MFMA rows.
// ------------------------------------------------------------------------------------------------
template <int FT>
__global__ __launch_bounds__(64 * 4 * NGRP) void time2_logits_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ xt,
                                                                     const bf16_t* __restrict__ wkT, int64_t ldw,
                                                                     float* __restrict__ slab, int rows, int heads,
                                                                     float scale) {
    constexpr int QPF = 16 / FT;           // queries per A fragment (rows = (query, frame))
    constexpr int NFR = 4 / QPF;           // fragments per wave (4 queries per wave); FT = 4: one fragment of 4 queries
    static_assert(QPF >= 1 && NFR >= 1, "FT in {4, 8, 16}");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = heads * CH, cc = blockIdx.x, nranges = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, wg = w & 3;
    char* sW = smem;
    char* sU = smem + CH * wrow_bytes(C) + grp * TQ * urow_bytes<16>(heads);
    const int urow = urow_bytes<16>(heads);
    const int ntiles = (rows + TQ - 1) / TQ;
    const int t_begin = (int)((int64_t)ntiles * blockIdx.y / nranges), t_end = (int)((int64_t)ntiles * (blockIdx.y + 1) / nranges);
    load_w_slice(sW, wkT, ldw, cc, C);
    __syncthreads();
    const int col = lane & 15, kg = lane >> 4;
    const int qi = col / FT, fl = col % FT;                       // this lane's A row = (query qi of the fragment, frame fl)
    const int hl = col < heads ? col : heads - 1;                 // head of this lane's B column (cols >= heads: duplicates)
    struct XFrag { bf16x8 v[NFR][2]; };
    auto load_x = [&](XFrag& x, int row0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const int64_t row = min(row0 + 4 * wg + QPF * i + qi, rows - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                x.v[i][ks] = *reinterpret_cast<const bf16x8*>(xt + (row * FT + fl) * C + cc * CH + 32 * ks + 8 * kg);
        }
    };
    const int niter = (t_end - t_begin + NGRP - 1) / NGRP;        // both groups run the same number of barriers
    // XD tiles of this group stay in flight (a CU needs ~50-100 KB in flight for its share of the HBM rate at 2-4 us
    // latency; a tile is 16 KB of x~ + 24 KB of q2)
    constexpr int XD = 3;
    XFrag xring[XD];
    QFrag qring[XD];
    // The ring loads are UNCONDITIONAL (rows past the end are clamped to the last row, tiles past the range re-read it):
    // with loads under an `if` hipcc cannot count how many are outstanding and drains s_waitcnt vmcnt(0) at every use,
    // which empties the ring each iteration (seen in the ISA: vmcnt(11) .. vmcnt(0) ladders).
#pragma unroll
    for (int k = 0; k < XD; ++k) {
        load_q_frags(qring[k], q2, (t_begin + grp + NGRP * k) * TQ, rows, C, heads, wg, lane);
        load_x(xring[k], (t_begin + grp + NGRP * k) * TQ);
    }
    for (int itb = 0; itb < niter; itb += XD) {
#pragma unroll
      for (int k = 0; k < XD; ++k) {
        const int it = itb + k;
        if (it >= niter) break;
        const int t = t_begin + NGRP * it + grp;
        const bool live = t < t_end;
        const int row0 = t * TQ;
        const XFrag xa = xring[k];
        const QFrag qa = qring[k];
        // q2 (19 MB, re-read by the 12 chunk owners: L2 / Infinity Cache, not "free") rides the same ring as x~
        load_q_frags(qring[k], q2, row0 + NGRP * XD * TQ, rows, C, heads, wg, lane);
        load_x(xring[k], row0 + NGRP * XD * TQ);
        if (live) compute_u_tile<16>(sW, sU, qa, C, heads, wg, lane);
        lds_barrier();
        if (live) {
#pragma unroll
            for (int i = 0; i < NFR; ++i) {
                f32x4 acc[QPF];
#pragma unroll
                for (int j = 0; j < QPF; ++j) {
                    const int sq = 4 * wg + QPF * i + j;          // query of the tile whose U is the B operand
                    acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8 ub = *reinterpret_cast<const bf16x8*>(sU + sq * urow + hl * (CH * 2 + 16) + (32 * ks + 8 * kg) * 2);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa.v[i][ks], ub, acc[j], 0, 0, 0);
                    }
                }
                // lane (col = head, kg): output rows 4 kg .. 4 kg + 3 = frames (4 kg) % FT .. of query (4 kg) / FT of the fragment
                const int jq = (4 * kg) / FT, f0 = (4 * kg) % FT;
                f32x4 v = acc[0];
#pragma unroll
                for (int j = 1; j < QPF; ++j)
                    if (jq == j) v = acc[j];
                const int row = row0 + 4 * wg + QPF * i + jq;
                if (row < rows && col < heads) {
                    float4 o = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
                    *reinterpret_cast<float4*>(slab + (((int64_t)cc * rows + row) * heads + col) * FT + f0) = o;
                }
            }
        }
        lds_barrier();                                           // U tile is free for the next tile
      }
    }
}

// forward 2/3: sum the chunk slabs, softmax over the frames -> attn2 [row][h][F]
template <int FT>
__global__ __launch_bounds__(256) void time2_softmax_kernel(const float* __restrict__ slab, float* __restrict__ attn2,
                                                            int64_t n, int nchunk) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one (row, h) per thread
    if (i >= n) return;
    float lg[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) lg[f] = 0.f;
#pragma unroll 4
    for (int c = 0; c < nchunk; ++c) {
        const float4* p = reinterpret_cast<const float4*>(slab + ((int64_t)c * n + i) * FT);
#pragma unroll
        for (int f4 = 0; f4 < FT / 4; ++f4) {
            const float4 v = p[f4];
            lg[4 * f4] += v.x; lg[4 * f4 + 1] += v.y; lg[4 * f4 + 2] += v.z; lg[4 * f4 + 3] += v.w;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int f = 0; f < FT; ++f) m = fmaxf(m, lg[f]);
    float den = 0.f;
#pragma unroll
    for (int f = 0; f < FT; ++f) { lg[f] = __expf(lg[f] - m); den += lg[f]; }
    const float inv = 1.f / den;
    float4* o = reinterpret_cast<float4*>(attn2 + i * FT);
#pragma unroll
    for (int f4 = 0; f4 < FT / 4; ++f4)
        o[f4] = make_float4(lg[4 * f4] * inv, lg[4 * f4 + 1] * inv, lg[4 * f4 + 2] * inv, lg[4 * f4 + 3] * inv);
}

// forward 3/3: out[s, (h,dd)] = sum_f attn2[s,h,f] x~[s,f,(h,dd)]; thread = (row, 8-channel group), 16-byte accesses
template <int FT>
__global__ __launch_bounds__(256) void time2_out_kernel(const bf16_t* __restrict__ xt, const float* __restrict__ attn2,
                                                        bf16_t* __restrict__ out, int64_t obs, int64_t ngroups, int S,
                                                        int heads) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool act = gi < ngroups;
    const int64_t g = act ? gi : ngroups - 1;
    const int gpr = heads * 8;
    const int row = (int)(g / gpr);
    const int cg = (int)(g - (int64_t)row * gpr);
    const int C = gpr * 8;
    uint4 xr[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) xr[f] = *reinterpret_cast<const uint4*>(xt + (((int64_t)row * FT + f) * C) + cg * 8);
    const float* arow = attn2 + ((int64_t)row * heads + (cg >> 3)) * FT;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const float a = arow[f];
        float xv[8];
        unpack8(xr[f], xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(a, xv[e], o[e]);
    }
    const int b = row / S, s = row - b * S;
    if (act) *reinterpret_cast<uint4*>(out + b * obs + (int64_t)s * C + cg * 8) = pack8(o);
}

// backward 1/2 (streaming, reads x~ once): block = RPB whole query rows, thread = (row, 8-channel group).
//   da[f]  = dout[s,h,:] . x~[s,f,h,:]  (8-lane reduction inside the head);  dl[f] = scale * a (da - sum_f a da)
//   dl of the row's heads meet in LDS, then  g[s,h,c8] = sum_f dl[s,f,h] x~[s,f,c8]  for every head h from the x~
//   registers the thread already holds -> g [row][h][C] bf16 (= d loss / d u), dl [row][f][16] bf16 for time2_dx_kernel.
template <int FT>
__global__ __launch_bounds__(256) void time2_dlg_kernel(const bf16_t* __restrict__ xt, const float* __restrict__ attn2,
                                                        const bf16_t* __restrict__ dout, int64_t dobs,
                                                        bf16_t* __restrict__ dl, bf16_t* __restrict__ gout, int rows,
                                                        int S, int HEADS, int RPB, float scale) {
    const int GPR = HEADS * 8, C = HEADS * CH;                     // 8-channel groups per row; RPB = 256 / GPR rows per block
    __shared__ float sdl[32][FT][MAXH];
    const int tid = threadIdx.x, rl = min(tid / GPR, RPB - 1), cg = tid - (tid / GPR) * GPR, h = cg >> 3;
    const int row_raw = blockIdx.x * RPB + rl;
    const bool act = row_raw < rows && tid < RPB * GPR;             // (threads past the last whole row of the block idle)
    const int row = act ? row_raw : rows - 1;
    const int b = row / S, s = row - b * S;
    uint4 xr[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) xr[f] = *reinterpret_cast<const uint4*>(xt + (((int64_t)row * FT + f) * C) + cg * 8);
    float gv[8];
    unpack8(*reinterpret_cast<const uint4*>(dout + b * dobs + (int64_t)s * C + cg * 8), gv);
    const float* arow = attn2 + ((int64_t)row * HEADS + h) * FT;
    float a[FT], da[FT], dot = 0.f;
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        a[f] = arow[f];
        float xv[8];
        unpack8(xr[f], xv);
        float p = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) p = fmaf(gv[e], xv[e], p);
        da[f] = sum8lanes(p);                                    // the head's 8 lanes are adjacent (GPR % 8 == 0)
        dot = fmaf(a[f], da[f], dot);
    }
    if (tid < RPB * GPR && (cg & 7) == 0) {
#pragma unroll
        for (int f = 0; f < FT; ++f) sdl[rl][f][h] = scale * a[f] * (da[f] - dot);
    }
    if (tid < RPB * GPR && cg < MAXH - HEADS) {                      // padded head columns: zeros (read by time2_dx_kernel)
#pragma unroll
        for (int f = 0; f < FT; ++f) sdl[rl][f][HEADS + cg] = 0.f;
    }
    __syncthreads();
    // dl row -> HBM (bf16 [row][f][16]): FT * 16 values per row, 8 per thread
    if (act) {
        for (int pc = cg; pc < FT * 2; pc += GPR) {
            const int f = pc >> 1, h0 = (pc & 1) * 8;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = sdl[rl][f][h0 + k];
            *reinterpret_cast<uint4*>(dl + ((int64_t)row * FT + f) * MAXH + h0) = pack8(v);
        }
    }
    // g[h][8 channels of this thread] for every head
    float xv[FT][8];
#pragma unroll
    for (int f = 0; f < FT; ++f) unpack8(xr[f], xv[f]);
    for (int hh = 0; hh < HEADS; ++hh) {
        float g8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const float d = sdl[rl][f][hh];
#pragma unroll
            for (int k = 0; k < 8; ++k) g8[k] = fmaf(d, xv[f][k], g8[k]);
        }
        if (act) *reinterpret_cast<uint4*>(gout + ((int64_t)row * HEADS + hh) * C + cg * 8) = pack8(g8);
    }
}

// ------------------------------------------------------------------------------------------------
// backward 2/2: dx~ for one channel chunk (recomputes U on chip like the forward; same two-group stream)
// ------------------------------------------------------------------------------------------------
template <int FT>
__global__ __launch_bounds__(64 * 4 * NGRP) void time2_dx_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ xt,
                                                                  const bf16_t* __restrict__ wkT, int64_t ldw,
                                                                  const float* __restrict__ attn2, const bf16_t* __restrict__ dl,
                                                                  const bf16_t* __restrict__ dout, int64_t dobs,
                                                                  bf16_t* __restrict__ dxt, int rows, int S, int heads) {
    constexpr int NTASK = FT / 2;          // dx~ tasks (query, frame, 8 channels) per thread: 16 * FT * 8 / 256
    constexpr int DLB = TQ * FT * MAXH * 2;    // bytes of a tile's dl block (bf16): 16 * FT * 16 * 2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = heads * CH, cc = blockIdx.x, nranges = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, wg = w & 3, gt = tid & 255;            // group, wave in group, thread in group
    char* sW = smem;
    char* sU = sW + CH * wrow_bytes(C) + grp * (TQ * urow_bytes<0>(heads) + DLB);
    bf16_t* sDL = reinterpret_cast<bf16_t*>(sU + TQ * urow_bytes<0>(heads));   // [16 s][FT][MAXH] dl (bf16)
    const int urow = urow_bytes<0>(heads);
    const int ntiles = (rows + TQ - 1) / TQ;
    const int t_begin = (int)((int64_t)ntiles * blockIdx.y / nranges), t_end = (int)((int64_t)ntiles * (blockIdx.y + 1) / nranges);
    load_w_slice(sW, wkT, ldw, cc, C);
    __syncthreads();
    // dx~ tasks: e = gt + 256 i -> 8-channel group e & 7, (query, frame) = e >> 3
    const int c8 = gt & 7;

    // every global operand of a tile (dout rows and a of the dx~ tasks, the dl block, q2 fragments) is loaded one tile
    // ahead, before the current tile's stores are issued (loads and stores retire in order on one counter)
    constexpr int NPC = DLB / 16, NDP = (NPC + 255) / 256;        // 16-byte pieces of the dl block; per thread
    struct In { uint4 dv[NTASK]; float av[NTASK]; uint4 dlp[NDP]; QFrag q; };
    auto load_in = [&](In& in, int row0) __attribute__((always_inline)) {
        const int npieces = (min(row0 + TQ, rows) - row0) * FT * 2;           // valid 16-byte pieces of the dl block
#pragma unroll
        for (int j = 0; j < NDP; ++j)
            in.dlp[j] = *reinterpret_cast<const uint4*>(dl + (int64_t)row0 * FT * MAXH + min(gt + 256 * j, npieces - 1) * 8);
        load_q_frags(in.q, q2, row0, rows, C, heads, wg, lane);
#pragma unroll
        for (int i = 0; i < NTASK; ++i) {
            const int sf = (gt + 256 * i) >> 3;
            const int row = min(row0 + sf / FT, rows - 1);
            const int b = row / S;
            in.av[i] = attn2[((int64_t)row * heads + cc) * FT + sf % FT];
            in.dv[i] = *reinterpret_cast<const uint4*>(dout + b * dobs + (int64_t)(row - b * S) * C + cc * CH + c8 * 8);
        }
    };
    const int niter = (t_end - t_begin + NGRP - 1) / NGRP;
    In nx;
    load_in(nx, min((t_begin + grp) * TQ, rows - 1));               // unconditional loads: see time2_logits_kernel
    for (int it = 0; it < niter; ++it) {
        const int t = t_begin + NGRP * it + grp;
        const bool live = t < t_end;
        const int row0 = t * TQ;
        const In in = nx;
        load_in(nx, min(row0 + NGRP * TQ, rows - 1));
        if (live) {
#pragma unroll
            for (int j = 0; j < NDP; ++j)
                if (gt + 256 * j < NPC) reinterpret_cast<uint4*>(sDL)[gt + 256 * j] = in.dlp[j];
            compute_u_tile<0>(sW, sU, in.q, C, heads, wg, lane);
        }
        lds_barrier();
        if (live) {
            // ---- dx~[s,f,c] = a[s,f,cc] dout[s,c] + sum_h dl[s,f,h] U[s,h,c] : task = (s, f, 8 channels) ----
#pragma unroll
            for (int i = 0; i < NTASK; ++i) {
                const int sf = (gt + 256 * i) >> 3, f = sf % FT, sq = sf / FT;
                const int row = row0 + sq;
                float dv[8], v[8], dlv[16];
                unpack8(in.dv[i], dv);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = in.av[i] * dv[k];
                const uint4* dlr = reinterpret_cast<const uint4*>(sDL + (sq * FT + f) * MAXH);
                unpack8(dlr[0], dlv);
                unpack8(dlr[1], dlv + 8);
#pragma unroll
                for (int h = 0; h < MAXH; ++h) {
                    if (h >= heads) break;
                    float uv[8];
                    unpack8(*reinterpret_cast<const uint4*>(sU + sq * urow + h * (CH * 2) + c8 * 16), uv);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaf(dlv[h], uv[k], v[k]);
                }
                if (row < rows) *reinterpret_cast<uint4*>(dxt + ((int64_t)row * FT + f) * C + cc * CH + c8 * 8) = pack8(v);
            }
        }
        lds_barrier();                                           // U / dl tiles are free for the next tile
    }
}

size_t time2_lds_fwd(int heads) { return (size_t)CH * wrow_bytes(heads * CH) + (size_t)NGRP * TQ * urow_bytes<16>(heads); }
size_t time2_lds_bwd(int heads, int F) {
    return (size_t)CH * wrow_bytes(heads * CH) + (size_t)NGRP * (TQ * urow_bytes<0>(heads) + TQ * F * MAXH * 2);
}

int ranges_for(int nchunk, int64_t rows) {
    int r = 256 / nchunk;                                          // one workgroup per CU
    const int64_t ntiles = (rows + TQ - 1) / TQ;
    if (r > ntiles) r = (int)ntiles;
    return r < 1 ? 1 : r;
}

}  // namespace

bool focus_traj_time2_ok(int F, int heads, int d, int dtype) {
    static const bool enabled = !(getenv("FOCUS_TIME2") && atoi(getenv("FOCUS_TIME2")) == 0);
    return enabled && dtype == FOCUS_BF16 && d == CH && heads >= 1 && heads <= MAXH && (F == 4 || F == 8 || F == 16) &&
           time2_lds_bwd(heads, F) <= 160 * 1024 && time2_lds_fwd(heads) <= 160 * 1024;
}

extern "C" size_t focus_traj_time2_workspace_bytes(int B, int S, int F, int heads, int d) {
    if (B <= 0 || S <= 0 || F <= 0 || heads <= 0 || d != CH) return 0;
    return (size_t)heads * B * heads * S * F * sizeof(float);       // one partial-logit slab per channel chunk
}

extern "C" int focus_traj_time2_fwd(const void* q2, const void* xt, const void* wkT, int64_t ldw, void* out,
                                    int64_t out_bstride, float* attn2, void* ws, size_t ws_bytes, int B, int S, int F,
                                    int heads, int d, int dtype, void* stream) {
    if (!q2 || !xt || !wkT || !out || !attn2 || !ws) return FOCUS_ERR_NULL;
    if (B <= 0 || S <= 0 || !focus_traj_time2_ok(F, heads, d, dtype)) return FOCUS_ERR_SHAPE;
    const int C = heads * CH;
    const int64_t rows = (int64_t)B * S;
    if (ldw < C || (ldw & 7) || (out_bstride & 7) || out_bstride < (int64_t)S * C || rows * F * C > 0x7fffffffLL * 8)
        return FOCUS_ERR_SHAPE;
    if (!focus_aligned(q2, 16) || !focus_aligned(xt, 16) || !focus_aligned(wkT, 16) || !focus_aligned(out, 16) ||
        !focus_aligned(ws, 16) || !focus_aligned(attn2, 16))
        return FOCUS_ERR_ALIGN;
    if (ws_bytes < focus_traj_time2_workspace_bytes(B, S, F, heads, d)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = C / CH;
    const size_t lds = time2_lds_fwd(heads);
    const float scale = 1.f / sqrtf((float)d);
    dim3 grid(nchunk, ranges_for(nchunk, rows));
#define TL(FT) do { \
        static bool once_##FT = (hipFuncSetAttribute((const void*)time2_logits_kernel<FT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
        (void)once_##FT; \
        hipLaunchKernelGGL((time2_logits_kernel<FT>), grid, dim3(64 * 4 * NGRP), lds, s, (const bf16_t*)q2, (const bf16_t*)xt, (const bf16_t*)wkT, ldw, (float*)ws, (int)rows, heads, scale); } while (0)
    if (F == 8) TL(8); else if (F == 4) TL(4); else TL(16);
#undef TL
    FOCUS_CHECK_LAUNCH();
    const int64_t n = rows * heads;
#define TS(FT) hipLaunchKernelGGL((time2_softmax_kernel<FT>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, (const float*)ws, attn2, n, nchunk)
    if (F == 8) TS(8); else if (F == 4) TS(4); else TS(16);
#undef TS
    FOCUS_CHECK_LAUNCH();
    const int64_t ng = rows * heads * 8;
    dim3 gv((unsigned)cdiv64(ng, 256));
#define TO(FT) hipLaunchKernelGGL((time2_out_kernel<FT>), gv, dim3(256), 0, s, (const bf16_t*)xt, attn2, (bf16_t*)out, out_bstride, ng, S, heads)
    if (F == 8) TO(8); else if (F == 4) TO(4); else TO(16);
#undef TO
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_traj_time2_bwd(const void* q2, const void* xt, const void* wkT, int64_t ldw, const float* attn2,
                                    const void* dout, int64_t dout_bstride, void* dxt, void* g, void* dl, int B, int S,
                                    int F, int heads, int d, int dtype, void* stream) {
    if (!q2 || !xt || !wkT || !attn2 || !dout || !dxt || !g || !dl) return FOCUS_ERR_NULL;
    if (B <= 0 || S <= 0 || !focus_traj_time2_ok(F, heads, d, dtype)) return FOCUS_ERR_SHAPE;
    const int C = heads * CH;
    const int64_t rows = (int64_t)B * S;
    if (ldw < C || (ldw & 7) || (dout_bstride & 7) || dout_bstride < (int64_t)S * C || rows * F * C > 0x7fffffffLL * 8)
        return FOCUS_ERR_SHAPE;
    if (!focus_aligned(q2, 16) || !focus_aligned(xt, 16) || !focus_aligned(wkT, 16) || !focus_aligned(dout, 16) ||
        !focus_aligned(dxt, 16) || !focus_aligned(g, 16) || !focus_aligned(dl, 16))
        return FOCUS_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = C / CH;
    const float scale = 1.f / sqrtf((float)d);
    {
        const int gpr = heads * 8, rpb = 256 / gpr;                // heads <= 16 -> at least 2 whole rows per 256-thread block
        dim3 gd((unsigned)cdiv64(rows, rpb));
#define TD(FT) hipLaunchKernelGGL((time2_dlg_kernel<FT>), gd, dim3(256), 0, s, (const bf16_t*)xt, attn2, (const bf16_t*)dout, dout_bstride, (bf16_t*)dl, (bf16_t*)g, (int)rows, S, heads, rpb, scale)
        if (F == 8) TD(8); else if (F == 4) TD(4); else TD(16);
#undef TD
        FOCUS_CHECK_LAUNCH();
    }
    const size_t lds = time2_lds_bwd(heads, F);
    dim3 grid(nchunk, ranges_for(nchunk, rows));
#define TB(FT) do { \
        static bool once_##FT = (hipFuncSetAttribute((const void*)time2_dx_kernel<FT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
        (void)once_##FT; \
        hipLaunchKernelGGL((time2_dx_kernel<FT>), grid, dim3(64 * 4 * NGRP), lds, s, (const bf16_t*)q2, (const bf16_t*)xt, (const bf16_t*)wkT, ldw, attn2, (const bf16_t*)dl, (const bf16_t*)dout, dout_bstride, (bf16_t*)dxt, (int)rows, S, heads); } while (0)
    if (F == 8) TB(8); else if (F == 4) TB(4); else TB(16);
#undef TB
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
