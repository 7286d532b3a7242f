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
// Decomposition (the same for forward and backward): workgroup = (channel chunk cc of 64 channels, query range), 8 waves,
// one workgroup per CU.  Its slice Wk^T[cc*64 .. +63][all (h,dd)] (96 KB bf16) stays in REGISTERS for the whole launch,
// split over the waves as MFMA A fragments (WFrag); per tile of 16 queries:
//     U^T[c, s]   = sum_dd WkT[c, h*64+dd] q2[s, h*64+dd]          MFMA 16x16x32, M = channels, N = queries, K = 64
//                   -> bf16 -> LDS image U[s][h][c] (double buffered, one barrier per tile)
//   forward  (time2_logits_kernel):
//     L[f, h]    += sum_c x~[s,f,c] U[s,h,c]   per query s            MFMA 16x16x32, M = frames, N = heads, K = 64;
//                   the A operand is read from x~ in HBM directly (16 B per lane, 128-B row segments); partial
//                   logits of the chunk go to slab[cc][row][h][f]; time2_softmax_kernel sums the C/64 slabs and takes
//                   the softmax over f (attn2 [B,S,h,F]); time2_out_kernel forms out = sum_f a x~ (reads x~ once more).
//   backward (time2_dlg_kernel, then time2_dx_kernel):
//     dl[s,f,h]   = scale a (da - sum_f a da),  da[s,f,h] = dout[s,h,:] . x~[s,f,h,:]
//     g[s,h,c]    = sum_f dl[s,f,h] x~[s,f,c]   ( = du )  -> HBM [h,B*S,C] bf16: the host forms
//                   dq2 = g[:,h,:] . Wk[h]^T  and  dWk[h] = q2[:,h,:]^T . g[:,h,:]  from it (two batched GEMMs).
//     dx~[s,f,c]  = a[s,f,h(c)] dout[s,c] + sum_h dl[s,f,h] U[s,h,c]     MFMA 16x16x32, M = channels, N = (query, frame),
//                   K = (query, head); U^T gathered from the LDS image by ds_read_b64_tr_b16.
// HBM per block and direction: x~ is read twice forward (logits, out) and once backward (dl + g), dx~ and g are
// written once; the k2 path moved k2/dk2 four more times and ran three 100352 x 768 x 768 GEMMs.
// Measured at the bench shape (B = 8, S = 1568, F = 8, 12 heads; rocprofv3 kernel trace): logits 72 us, softmax 11, out 27
// | dlg 68, dx 101, dq2 GEMM 82, dWk GEMM 75 -- forward 156 vs 264 us, backward 347 vs 445 us for the k2 path, bench
// step 35.2 vs 38.0 ms.
#include "focus_common.h"
#include <cstdlib>
#include "traj_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int CH = 64;                 // channels per chunk (= head dim: chunk cc holds exactly head cc's channels)
constexpr int TQ = 16;                 // queries per tile (MFMA N)
constexpr int MAXH = 16;               // heads <= 16 (MFMA N of the logit product)

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
// hipcc drains s_waitcnt vmcnt(0): that would wait for the prefetch ring (and this tile's stores) at every barrier and
// serialise the stream.  The tiles only hand LDS data between waves.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Which (channel chunk, query range) a workgroup owns: 256 workgroups of 512 threads, one per CU.  Every owner of a range
// reads that range's q2 (and, backward, its attn2 / dl / dout), and an XCD's L2 only shares those reads among the owners
// it hosts: spread over all 8 XCDs the logits kernel fetched 332 MB against 173 algorithmic, the dx kernel 237 against 46
// (profiles/r02_pmc_hbm_traffic.txt).  The nchunk owners are therefore dealt out in groups of g (the largest divisor of
// nchunk <= 8: 6 for 12 chunks) that sit on ONE XCD -- workgroup ids go to the XCDs round-robin (id % 8), so XCD x holds
// ids x, x + 8, ...: its 32 slots = floor(32 / g) groups -- and the nchunk / g groups of a range land on neighbouring
// XCDs: a range is fetched by 2 XCDs instead of 8.  (All 12 owners on one XCD would idle 8 of its 32 CUs: measured
// slower.)  12 chunks: 40 groups = 20 ranges, 16 of the 256 workgroups idle.  FOCUS_T2_SPREAD=1: id -> (id % nchunk,
// id / nchunk), the owners of a range on all XCDs (kept for A/B).
struct Owner { int cc, range, nranges; };
__device__ __forceinline__ Owner owner_of_block(int nchunk, int spread) {
    Owner o;
    const int id = blockIdx.x;
    if (spread) {
        o.nranges = gridDim.x / nchunk;
        o.cc = id % nchunk;
        o.range = id < o.nranges * nchunk ? id / nchunk : -1;
        return o;
    }
    int g = 1;
    for (int d = 2; d <= 8; ++d)
        if (nchunk % d == 0) g = d;
    const int xcd = id & 7, j = id >> 3, per_xcd = gridDim.x >> 3;
    const int gpx = per_xcd / g, parts = nchunk / g;
    const int gi = xcd + 8 * (j / g);                              // group index; groups parts*r .. parts*r + parts-1 = range r
    o.nranges = 8 * gpx / parts;
    o.cc = (gi % parts) * g + j % g;
    o.range = (j < gpx * g && gi < o.nranges * parts) ? gi / parts : -1;
    return o;
}

// A block's tile range in whole turns of its prefetch ring (DEPTH tiles): the unrolled ring loops below then have one exit,
// at the loop head, and no register of the ring is live across a second one (a mid-turn exit made hipcc rotate ring
// registers with v_mov on the back edge -- copies of loads still in flight; focus_amd/build.py lint_hand_loads).  The last
// range may run up to DEPTH - 1 tiles past ntiles: their loads clamp to the last row, their stores are masked by row < rows.
__device__ __forceinline__ int turn_begin(int ntiles, int range, int nranges, int depth) {
    const int turns = (ntiles + depth - 1) / depth;
    return (int)((int64_t)turns * range / nranges) * depth;
}

// Global loads the compiler does not track (hand-counted s_waitcnt vmcnt, cdna_hip_programming.md 5.7 form (iii)): used
// for the prefetch rings -- hipcc's own counting of ring loads in these loops ends in vmcnt(0) ladders.
__device__ __forceinline__ void gload16_asm(bf16x8& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload4_asm(float& dst, const void* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void pin(bf16x8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
// Leaving a slot: whatever is computed FROM a ring slot must itself be pinned before the slot is refilled.  Volatile asm
// statements keep their order, plain VALU work does not: hipcc once sank the residual add of time2_dx_lds_kernel below the
// refill, which made the old and the new contents of the slot live together, gave the refill other registers and rotated
// them back with v_mov on the loop's back edge -- copies of loads still in flight (the build's lint_hand_loads reports it).
__device__ __forceinline__ void pin(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// ---- U images in LDS (conflict degrees from tools/lds_bank_model.py, lane groups of MI355X_MICROARCH.md "LDS") ----
// forward : U[16 s][heads][64 c], head pitch 128 + 16 B, query pitch + 16 B: read as the MFMA B operand with the HEAD on
//           the lane (ds_read_b128, 2-way), written with the query on the lane (ds_write_b64, conflict-free).
// backward: U[16 s][16 h][64 c], head pitch 128 + 8 B, the 8-byte chunks of a row XOR-ed with (s >> 1) & 7: gathered
//           h-major by ds_read_b64_tr_b16 (conflict-free), written 2-way.  Head rows >= heads are zero.
constexpr int FWD_HP = CH * 2 + 16;
__host__ __device__ __forceinline__ int fwd_urow(int heads) { return heads * FWD_HP + 16; }
constexpr int BWD_HP = CH * 2 + 8, BWD_UROW = MAXH * BWD_HP;
__device__ __forceinline__ int bwd_off(int s, int h, int chunk) { return s * BWD_UROW + h * BWD_HP + ((chunk ^ ((s >> 1) & 7)) << 3); }

// ---- the chunk's weights stay in REGISTERS ----
// Wk^T[cc*64 .. +63][all (h,dd)] is 96 KB: as an LDS image every 16-query tile re-read all of it (the LDS array, not
// HBM, bounded the kernels: 2070 of 8200 cycles per tile in the U product, the rest barrier waits on it).  Split into
// units (head h, 32-channel half) it is 2 heads x 8 waves ... : wave w holds units w*UPW .. w*UPW+UPW-1 as MFMA A
// fragments (16 VGPRs per unit), UPW = ceil(2 heads / 8) <= 4, and LDS carries only the U tiles.
template <int UPW> struct WFrag { bf16x8 v[UPW][2][2]; };             // [unit][16-channel tile of the half][k step]
template <int UPW> struct NHeads { static constexpr int value = UPW == 3 ? 2 : (UPW + 1) / 2; };   // distinct heads of a wave
template <int NHW> struct QFrag { bf16x8 v[NHW][2]; };

template <int UPW>
__device__ __forceinline__ void load_w_frags(WFrag<UPW>& wf, const bf16_t* __restrict__ wkT, int64_t ldw, int cc, int heads,
                                             int w, int lane) {
    const int col = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = min(w * UPW + k, 2 * heads - 1), h = u >> 1, half = u & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf.v[k][m][ks] = *reinterpret_cast<const bf16x8*>(wkT + (int64_t)(cc * CH + 32 * half + 16 * m + col) * ldw +
                                                                 h * CH + 32 * ks + 8 * kg);
    }
}

// q2 fragments (MFMA B operand: column = query, k = dd) of this wave's heads for the tile starting at row0 (ring load)
template <int UPW>
__device__ __forceinline__ void load_q_frags(QFrag<NHeads<UPW>::value>& q, const bf16_t* __restrict__ q2, int row0, int rows,
                                             int C, int heads, int w, int lane) {
    const int col = lane & 15, kg = lane >> 4, hfirst = (w * UPW) >> 1;
    const int64_t qoff = (int64_t)min(row0 + col, rows - 1) * C;  // this lane's query (MFMA column), clamped
#pragma unroll
    for (int i = 0; i < NHeads<UPW>::value; ++i) {
        const int h = min(hfirst + i, heads - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) gload16_asm(q.v[i][ks], q2 + qoff + h * CH + 32 * ks + 8 * kg);
    }
}

// U[s][h][c] of one 16-query tile for the units of this wave: U^T = WkT_unit . q2_h^T  (M = channels, N = queries, K = dd)
template <int UPW, bool BWD>
__device__ __forceinline__ void compute_u_tile(const WFrag<UPW>& wf, const QFrag<NHeads<UPW>::value>& q, char* sU, int urow,
                                               int heads, int w, int lane) {
    const int col = lane & 15, kg = lane >> 4, hfirst = (w * UPW) >> 1;
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = w * UPW + k;
        if (u >= 2 * heads) break;
        const int h = u >> 1, half = u & 1;
        bf16x8 qa[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (NHeads<UPW>::value == 1) qa[ks] = q.v[0][ks];
            else qa[ks] = h != hfirst ? q.v[NHeads<UPW>::value - 1][ks] : q.v[0][ks];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.v[k][m][0], qa[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.v[k][m][1], qa[1], acc, 0, 0, 0);
            // acc[r] = U^T[c = 32 half + 16 m + 4 kg + r][s = col]
            Pk4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p.e[r] = f32_to_bf16(acc[r]);
            const int mt = 2 * half + m;
            if (BWD) *reinterpret_cast<uint2*>(sU + bwd_off(col, h, 4 * mt + kg)) = p.u;
            else *reinterpret_cast<uint2*>(sU + col * urow + h * FWD_HP + (16 * mt + 4 * kg) * 2) = p.u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward 1/3: partial logits of one channel chunk.  512 threads = 8 waves, one workgroup per CU, every wave on every
// tile: U units (above) -> LDS (double buffered: one barrier per tile) -> wave w contracts the x~ fragment(s) of
// 16 / F queries against their U rows.  x~ is the MFMA A operand straight from HBM (16 B per lane).
// ------------------------------------------------------------------------------------------------
template <int FT, int UPW>
__global__ __launch_bounds__(512) void time2_logits_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ xt,
                                                           const bf16_t* __restrict__ wkT, int64_t ldw,
                                                           float* __restrict__ slab, int rows, int heads, float scale, int spread) {
    constexpr int QPF = 16 / FT;           // queries per A fragment (rows = (query, frame))
    constexpr int NFR = FT > 8 ? FT / 8 : 1;   // fragments per wave: a tile has FT fragments (F = 4: waves 4..7 have none)
    constexpr int NHW = NHeads<UPW>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Owner own = owner_of_block(heads, spread);
    if (own.range < 0) return;
    const int C = heads * CH, cc = own.cc;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int urow = fwd_urow(heads);
    const int ntiles = (rows + TQ - 1) / TQ;
    const int t_begin = turn_begin(ntiles, own.range, own.nranges, 3), t_end = turn_begin(ntiles, own.range + 1, own.nranges, 3);   // 3 = XD below
    const int niter = t_end - t_begin;
    WFrag<UPW> wf;
    load_w_frags<UPW>(wf, wkT, ldw, cc, heads, w, lane);
    const int col = lane & 15, kg = lane >> 4;
    const int qi = col / FT, fl = col % FT;                       // this lane's A row = (query qi of the fragment, frame fl)
    const int hl = col < heads ? col : heads - 1;                 // head of this lane's B column (cols >= heads: duplicates)
    // Software pipeline: between two barriers a wave forms its U units of tile t + 1 (buffer (it+1) & 1) AND contracts its
    // x~ fragment of tile t against buffer it & 1 -- two independent chains (MFMA + LDS write | LDS read + MFMA + store)
    // for the scheduler to overlap; done one after the other with every wave in step the tile time was their sum.
    // Register rings of XD tiles (a CU needs ~50-100 KB in flight for its share of the HBM rate at 2-4 us of latency; a
    // tile is 16 KB of x~ + 24..32 KB of q2 fragments): q2 of tile t + 1 and x~ of tile t are consumed together, so the q
    // ring runs one tile ahead of the x ring.  The ring loads are inline-asm loads with a hand-counted wait: the issue
    // order is  q0 x0 q1 x1 q2 x2 | q0' | (q1' x0') (q2' x1') ...  with exactly NQ + NX = LPT loads per tile and wave
    // (unconditional: rows past the end are clamped), and the loads a tile consumes are always followed by 2 LPT younger
    // ones: s_waitcnt vmcnt((XD-1) * LPT); the stores issued in between only make the wait stricter.
    constexpr int XD = 3, NX = NFR * 2, LPT = NHW * 2 + NX;
    struct XFrag { bf16x8 v[NFR][2]; };
    auto load_x = [&](XFrag& x, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const int fi = min(w + 8 * i, FT - 1);
            const int64_t row = min(t * TQ + QPF * fi + qi, rows - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) gload16_asm(x.v[i][ks], xt + (row * FT + fl) * C + cc * CH + 32 * ks + 8 * kg);
        }
    };
    auto pin_q = [&](QFrag<NHW>& q) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NHW; ++i) { pin(q.v[i][0]); pin(q.v[i][1]); }
    };
    auto pin_x = [&](XFrag& x) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NFR; ++i) { pin(x.v[i][0]); pin(x.v[i][1]); }
    };
    QFrag<NHW> qr[XD];
    XFrag xr[XD];
#pragma unroll
    for (int k = 0; k < XD; ++k) {
        load_q_frags<UPW>(qr[k], q2, (t_begin + k) * TQ, rows, C, heads, w, lane);
        load_x(xr[k], t_begin + k);
    }
    if (niter > 0) {                                              // U of the first tile
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 1) * LPT + NX) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pin_q(qr[0]);
        compute_u_tile<UPW, false>(wf, qr[0], smem, urow, heads, w, lane);
        asm volatile("" ::: "memory");
        pin_q(qr[0]);
        load_q_frags<UPW>(qr[0], q2, (t_begin + XD) * TQ, rows, C, heads, w, lane);
        lds_barrier();
    }
    for (int itb = 0; itb < niter; itb += XD) {
#pragma unroll
      for (int k = 0; k < XD; ++k) {
        const int k1 = (k + 1) % XD;                              // (static after unrolling)
        const int it = itb + k;
        const int t = t_begin + it, row0 = t * TQ;
        const char* sU = smem + (it & 1) * (TQ * urow);
        char* sUn = smem + ((it + 1) & 1) * (TQ * urow);
        // the contraction's B operands first: their LDS latency passes under the U product
        bf16x8 ub[NFR][QPF][2];
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const int fi = min(w + 8 * i, FT - 1);
#pragma unroll
            for (int j = 0; j < QPF; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    ub[i][j][ks] = *reinterpret_cast<const bf16x8*>(sU + (QPF * fi + j) * urow + hl * FWD_HP + (32 * ks + 8 * kg) * 2);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 1) * LPT) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pin_q(qr[k1]);
        pin_x(xr[k]);
        if (it + 1 < niter) compute_u_tile<UPW, false>(wf, qr[k1], sUn, urow, heads, w, lane);
        f32x4 res[NFR];
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            f32x4 acc[QPF];
#pragma unroll
            for (int j = 0; j < QPF; ++j) {                       // query QPF fi + j of the tile: its U is the B operand
                acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xr[k].v[i][ks], ub[i][j][ks], acc[j], 0, 0, 0);
            }
            // lane (col = head, kg): output rows 4 kg .. 4 kg + 3 = frames (4 kg) % FT .. of query (4 kg) / FT of the fragment
            res[i] = acc[0];
#pragma unroll
            for (int j = 1; j < QPF; ++j)
                if ((4 * kg) / FT == j) res[i] = acc[j];
        }
        // q slot k1 and x slot k are consumed (their MFMAs have been issued): refill them, then store
        asm volatile("" ::: "memory");
        pin_q(qr[k1]);
        pin_x(xr[k]);
        load_q_frags<UPW>(qr[k1], q2, (t + 1 + XD) * TQ, rows, C, heads, w, lane);
        load_x(xr[k], t + XD);
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const int fi = w + 8 * i;
            const int jq = (4 * kg) / FT, f0 = (4 * kg) % FT;
            const int row = row0 + QPF * fi + jq;
            if (fi < FT && row < rows && col < heads) {
                const f32x4 v = res[i];
                float4 o = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
                *reinterpret_cast<float4*>(slab + (((int64_t)cc * rows + row) * heads + col) * FT + f0) = o;
            }
        }
        lds_barrier();                                           // one barrier per tile: U(t+1) complete, U(t) free
      }
    }
    // the ring loads of the tiles past the end are still in flight and hipcc does not know: drain them before anything
    // below may reuse their destination registers (an address register overwritten by a late load faults)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// forward 1/3, LDS-staged operands.  The kernel above loads q2 / x~ straight into MFMA operand layout: a 16-byte-per-lane
// load then touches 16 rows x 64 B, and the L1/TA serves that pattern at ~16 B/clk/CU from L2 against ~32 B/clk/CU for
// row-contiguous loads (tools/probe/ta_pattern_probe.hip: 32 vs 64 GB/s per CU) -- with 48 KB per tile that rate, not HBM
// and not the matrix pipe, set the tile time (1.6 us per tile whether or not any MFMA ran).  Here every thread loads
// row-contiguous 16-byte pieces (q2: whole rows, each element ONCE instead of once per head group; x~: 128-byte chunk
// rows), two tiles ahead in registers (inline-asm loads, counted waits), stores them to padded LDS images and the
// fragments are read back with ds_read_b128 (pitches + 16 B: conflict-free for the 16-lane groups of both reads).
// Per tile: A  fragments of x~(t), U(t), q2(t+1) -> registers | barrier | W  registers (tile t+2) -> the stage x~(t) left
//           | C  issue loads of tile t+4 | B  U(t+1), contraction(t), store | barrier.
// LDS: 2 U buffers + 2 stages of (q2 tile, x~ tile): 139 KB for 12 heads, F = 8; larger shapes take the direct kernel.
// ------------------------------------------------------------------------------------------------
constexpr int XP = CH * 2 + 16;                                    // x~ chunk row pitch in LDS
__host__ __device__ __forceinline__ int q_pitch(int heads) { return heads * CH * 2 + 16; }
__host__ __device__ __forceinline__ size_t time2_lds_staged(int heads, int F) {
    return (size_t)2 * TQ * fwd_urow(heads) + (size_t)2 * TQ * q_pitch(heads) + (size_t)2 * TQ * F * XP;
}

template <int FT, int UPW>
__global__ __launch_bounds__(512) void time2_logits_lds_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ xt,
                                                               const bf16_t* __restrict__ wkT, int64_t ldw,
                                                               float* __restrict__ slab, int rows, int heads, float scale, int spread) {
    constexpr int QPF = 16 / FT, NFR = FT > 8 ? FT / 8 : 1, NHW = NHeads<UPW>::value;
    constexpr int XROWS = TQ * FT;                                 // (query, frame) rows of a tile
    constexpr int NXP = XROWS * 8 / 512 > 0 ? XROWS * 8 / 512 : 1; // 16-byte x~ pieces per thread
    constexpr int NQP = 4;                                         // 16-byte q2 pieces per thread (16 rows x C/8 <= 2048)
    constexpr int LG = NQP + NXP;                                  // loads per thread and tile (fixed: counted waits)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Owner own = owner_of_block(heads, spread);
    if (own.range < 0) return;
    const int C = heads * CH, cc = own.cc;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int urow = fwd_urow(heads), QP = q_pitch(heads);
    char* sUb = smem;
    char* sQb = smem + 2 * TQ * urow;
    char* sXb = sQb + 2 * TQ * QP;
    const int ntiles = (rows + TQ - 1) / TQ;
    const int t_begin = turn_begin(ntiles, own.range, own.nranges, 2), t_end = turn_begin(ntiles, own.range + 1, own.nranges, 2);
    const int niter = t_end - t_begin;
    if (niter <= 0) return;
    WFrag<UPW> wf;
    load_w_frags<UPW>(wf, wkT, ldw, cc, heads, w, lane);
    const int col = lane & 15, kg = lane >> 4;
    const int hl = col < heads ? col : heads - 1, hfirst = (w * UPW) >> 1;
    // this thread's staging pieces (fixed for the whole launch)
    const int qpr = C / 8, nqp = TQ * qpr;                         // 16-byte pieces per q2 row / per q2 tile
    int qrow[NQP], qcol[NQP], qlds[NQP];
#pragma unroll
    for (int i = 0; i < NQP; ++i) {
        const int p = tid + 512 * i, pc = min(p, nqp - 1);
        qrow[i] = pc / qpr;
        qcol[i] = (pc - qrow[i] * qpr) * 8;
        qlds[i] = p < nqp ? qrow[i] * QP + qcol[i] * 2 : -1;
    }
    int xq[NXP], xoff[NXP], xlds[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int p = tid + 512 * i, pc = min(p, XROWS * 8 - 1), r = pc >> 3, c16 = pc & 7;
        xq[i] = r / FT;
        xoff[i] = (r % FT) * C + cc * CH + c16 * 8;
        xlds[i] = p < XROWS * 8 ? r * XP + c16 * 16 : -1;
    }
    struct Pre { bf16x8 q[NQP]; bf16x8 x[NXP]; };
    auto issue = [&](Pre& pr, int t) __attribute__((always_inline)) {
        const int row0 = t * TQ;
#pragma unroll
        for (int i = 0; i < NQP; ++i) gload16_asm(pr.q[i], q2 + (int64_t)min(row0 + qrow[i], rows - 1) * C + qcol[i]);
#pragma unroll
        for (int i = 0; i < NXP; ++i) gload16_asm(pr.x[i], xt + (int64_t)min(row0 + xq[i], rows - 1) * FT * C + xoff[i]);
    };
    auto pin_pre = [&](Pre& pr) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NQP; ++i) pin(pr.q[i]);
#pragma unroll
        for (int i = 0; i < NXP; ++i) pin(pr.x[i]);
    };
    auto stash = [&](const Pre& pr, int stage) __attribute__((always_inline)) {
        char* sq = sQb + stage * (TQ * QP);
        char* sx = sXb + stage * (XROWS * XP);
#pragma unroll
        for (int i = 0; i < NQP; ++i)
            if (qlds[i] >= 0) *reinterpret_cast<bf16x8*>(sq + qlds[i]) = pr.q[i];
#pragma unroll
        for (int i = 0; i < NXP; ++i)
            if (xlds[i] >= 0) *reinterpret_cast<bf16x8*>(sx + xlds[i]) = pr.x[i];
    };
    auto read_q = [&](QFrag<NHW>& qf, int stage) __attribute__((always_inline)) {
        const char* sq = sQb + stage * (TQ * QP) + col * QP;
#pragma unroll
        for (int i = 0; i < NHW; ++i) {
            const int h = min(hfirst + i, heads - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf.v[i][ks] = *reinterpret_cast<const bf16x8*>(sq + (h * CH + 32 * ks + 8 * kg) * 2);
        }
    };
    Pre pre[2];
    // prologue: tiles 0, 1 -> LDS stages 0, 1; tiles 2, 3 -> registers; U(0)
    issue(pre[0], t_begin);
    issue(pre[1], t_begin + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin_pre(pre[0]); pin_pre(pre[1]);
    stash(pre[0], 0);
    stash(pre[1], 1);
    asm volatile("" ::: "memory");
    pin_pre(pre[0]); pin_pre(pre[1]);
    issue(pre[0], t_begin + 2);
    issue(pre[1], t_begin + 3);
    lds_barrier();
    {
        QFrag<NHW> qf;
        read_q(qf, 0);
        compute_u_tile<UPW, false>(wf, qf, sUb, urow, heads, w, lane);
    }
    lds_barrier();
    for (int itb = 0; itb < niter; itb += 2) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int it = itb + k;
        const int t = t_begin + it, row0 = t * TQ;
        // A: this tile's x~ fragments and U rows, the next tile's q2 fragments -> registers
        const char* sx = sXb + k * (XROWS * XP);
        const char* sU = sUb + k * (TQ * urow);
        bf16x8 xf[NFR][2], ub[NFR][QPF][2];
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const int fi = min(w + 8 * i, FT - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[i][ks] = *reinterpret_cast<const bf16x8*>(sx + (16 * fi + col) * XP + (32 * ks + 8 * kg) * 2);
#pragma unroll
            for (int j = 0; j < QPF; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    ub[i][j][ks] = *reinterpret_cast<const bf16x8*>(sU + (QPF * fi + j) * urow + hl * FWD_HP + (32 * ks + 8 * kg) * 2);
        }
        QFrag<NHW> qf;
        read_q(qf, k ^ 1);
        lds_barrier();                                           // every wave has what it needs of stage k
        // W: tile it + 2 (loaded two iterations ago) -> stage k;  C: its register slot re-issued for tile it + 4
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LG) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pin_pre(pre[k]);
        stash(pre[k], k);
        asm volatile("" ::: "memory");
        pin_pre(pre[k]);
        issue(pre[k], t + 4);
        // B: U of the next tile, contraction of this one
        if (it + 1 < niter) compute_u_tile<UPW, false>(wf, qf, sUb + (k ^ 1) * (TQ * urow), urow, heads, w, lane);
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            f32x4 acc[QPF];
#pragma unroll
            for (int j = 0; j < QPF; ++j) {
                acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i][ks], ub[i][j][ks], acc[j], 0, 0, 0);
            }
            f32x4 v = acc[0];
#pragma unroll
            for (int j = 1; j < QPF; ++j)
                if ((4 * kg) / FT == j) v = acc[j];
            const int fi = w + 8 * i;
            const int jq = (4 * kg) / FT, f0 = (4 * kg) % FT;
            const int row = row0 + QPF * fi + jq;
            if (fi < FT && row < rows && col < heads) {
                float4 o = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
                *reinterpret_cast<float4*>(slab + (((int64_t)cc * rows + row) * heads + col) * FT + f0) = o;
            }
        }
        lds_barrier();                                           // U(t+1) and stage k (tile it + 2) are complete
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // late ring loads: see time2_logits_kernel
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
//   registers the thread already holds -> g [h][row][C] bf16 (= d loss / d u; head-major, so that each head's slice is a
//   dense matrix for the two GEMMs that consume it), dl [row][f][16] bf16 for time2_dx_kernel.
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
        if (act) *reinterpret_cast<uint4*>(gout + ((int64_t)hh * rows + row) * C + cg * 8) = pack8(g8);
    }
}

// ------------------------------------------------------------------------------------------------
// backward 2/2: dx~ for one channel chunk.  U is recomputed on chip like the forward; the contraction over the heads
//     dx~[s,f,c] = a[s,f,h(c)] dout[s,c] + sum_h dl[s,f,h] U[s,h,c]
// is an MFMA with M = channels, N = (query, frame) columns, K = (query j' of the column tile, head): the B operand is
// dl masked to its own query (block diagonal, read from HBM in operand layout), the A operand is U^T gathered h-major
// from the [s][h][c] image by ds_read_b64_tr_b16.  M row 4 p + e of tile mt is channel 16 p + 4 mt + e, so a lane ends
// with 16 consecutive channels of one (query, frame) row: two 16-byte stores, two 16-byte dout loads.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
union TrFrag { bf16x8 v; s16x4 t[2]; };

template <int FT, int UPW>
__global__ __launch_bounds__(512) void time2_dx_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ wkT, int64_t ldw,
                                                       const float* __restrict__ attn2, const bf16_t* __restrict__ dl,
                                                       const bf16_t* __restrict__ dout, int64_t dobs,
                                                       bf16_t* __restrict__ dxt, int rows, int S, int heads, int spread) {
    constexpr int QPF = 16 / FT;           // queries per column tile (columns = (query, frame))
    constexpr int NNT = FT > 8 ? FT / 8 : 1;   // column tiles per wave: a tile of 16 queries has FT of them
    constexpr int KS = QPF > 2 ? QPF / 2 : 1;  // k steps: K = QPF queries x 16 heads (F = 16: the upper 16 slots are zero)
    constexpr int NHW = NHeads<UPW>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Owner own = owner_of_block(heads, spread);
    if (own.range < 0) return;
    const int C = heads * CH, cc = own.cc;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (rows + TQ - 1) / TQ;
    const int t_begin = turn_begin(ntiles, own.range, own.nranges, FT > 8 ? 2 : 3), t_end = turn_begin(ntiles, own.range + 1, own.nranges, FT > 8 ? 2 : 3);   // = XD below
    const int niter = t_end - t_begin;
    WFrag<UPW> wf;
    load_w_frags<UPW>(wf, wkT, ldw, cc, heads, w, lane);
    for (int e = tid; e < 2 * TQ * BWD_UROW / 16; e += 512)         // head rows >= heads stay zero (their dl is zero, but
        reinterpret_cast<uint4*>(smem)[e] = make_uint4(0, 0, 0, 0);   // 0 x garbage must not be NaN)
    __syncthreads();
    const int col = lane & 15, kg = lane >> 4;
    const int j = col / FT, f = col % FT;                          // this lane's output column = (query j of the tile, frame f)
    const int hb = QPF >= 2 ? 8 * (kg >> 1) : 8 * (kg & 1);        // heads hb .. hb + 7 in this lane group's k slots
    const int tq = (lane & 15) >> 2, tp = lane & 3;                // transposed read: this lane supplies head row tq, channels 16 tp ..

    // pipeline and rings as in time2_logits_kernel: U of tile t + 1 and the dx~ product of tile t share a barrier interval
    struct DFrag { bf16x8 dl[NNT]; float a[NNT]; bf16x8 dout[NNT][2]; };
    constexpr int XD = FT > 8 ? 2 : 3, NX = NNT * 4, LPT = NHW * 2 + NX;   // (F = 16: a third slot would spill, and a
                                                                         // spilled ring register is stored before its load lands)
    auto load_d = [&](DFrag& x, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            const int nt = min(w + 8 * i, FT - 1);
            const int row = min(t * TQ + QPF * nt + j, rows - 1);
            const int b = row / S;
            gload16_asm(x.dl[i], dl + ((int64_t)row * FT + f) * MAXH + hb);
            gload4_asm(x.a[i], attn2 + ((int64_t)row * heads + cc) * FT + f);
            const bf16_t* dp = dout + b * dobs + (int64_t)(row - b * S) * C + cc * CH + 16 * kg;
            gload16_asm(x.dout[i][0], dp);
            gload16_asm(x.dout[i][1], dp + 8);
        }
    };
    auto pin_q = [&](QFrag<NHW>& q) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NHW; ++i) { pin(q.v[i][0]); pin(q.v[i][1]); }
    };
    auto pin_d = [&](DFrag& x) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NNT; ++i) { pin(x.dl[i]); pin(x.a[i]); pin(x.dout[i][0]); pin(x.dout[i][1]); }
    };
    QFrag<NHW> qr[XD];
    DFrag dr[XD];
#pragma unroll
    for (int k = 0; k < XD; ++k) {
        load_q_frags<UPW>(qr[k], q2, (t_begin + k) * TQ, rows, C, heads, w, lane);
        load_d(dr[k], t_begin + k);
    }
    if (niter > 0) {                                              // U of the first tile
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 1) * LPT + NX) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pin_q(qr[0]);
        compute_u_tile<UPW, true>(wf, qr[0], smem, 0, heads, w, lane);
        asm volatile("" ::: "memory");
        pin_q(qr[0]);
        load_q_frags<UPW>(qr[0], q2, (t_begin + XD) * TQ, rows, C, heads, w, lane);
        lds_barrier();
    }
    for (int itb = 0; itb < niter; itb += XD) {
#pragma unroll
      for (int k = 0; k < XD; ++k) {
        const int k1 = (k + 1) % XD;                              // (static after unrolling)
        const int it = itb + k;
        const int t = t_begin + it, row0 = t * TQ;
        const char* sU = smem + (it & 1) * (TQ * BWD_UROW);
        char* sUn = smem + ((it + 1) & 1) * (TQ * BWD_UROW);
        // the A operands (U^T of this tile, gathered h-major) first: their LDS latency passes under the U product
        TrFrag af[NNT][KS][4];
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            const int nt = min(w + 8 * i, FT - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int jp = QPF >= 2 ? (kg & 1) + 2 * ks : 0;  // query of this lane group's k slots
                const int sa = QPF * nt + jp;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    af[i][ks][mt].t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sU + bwd_off(sa, hb + tq, 4 * tp + mt)));
                    af[i][ks][mt].t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sU + bwd_off(sa, hb + 4 + tq, 4 * tp + mt)));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 1) * LPT) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pin_q(qr[k1]);
        pin_d(dr[k]);
        if (it + 1 < niter) compute_u_tile<UPW, true>(wf, qr[k1], sUn, 0, heads, w, lane);
        uint4 o[NNT][2];
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int jp = QPF >= 2 ? (kg & 1) + 2 * ks : 0;
                const bool keep = QPF >= 2 ? jp == j : kg < 2;
                bf16x8 bfr = dr[k].dl[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) bfr[e] = keep ? bfr[e] : (__bf16)0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks][mt].v, bfr, acc[mt], 0, 0, 0);
            }
            // lane (col = (j, f), kg): acc[mt][r] = channel 16 kg + 4 mt + r of row (j, f)
            float dv[16], v[16];
            unpack8(*reinterpret_cast<const uint4*>(&dr[k].dout[i][0]), dv);
            unpack8(*reinterpret_cast<const uint4*>(&dr[k].dout[i][1]), dv + 8);
            const float av = dr[k].a[i];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * mt + r] = fmaf(av, dv[4 * mt + r], acc[mt][r]);
            o[i][0] = pack8(v);
            o[i][1] = pack8(v + 8);
            pin(o[i][0]); pin(o[i][1]);                          // the last use of this slot's operands is BEFORE its refill:
        }                                                        // see "leaving a slot" at pin()
        asm volatile("" ::: "memory");
        pin_q(qr[k1]);
        pin_d(dr[k]);
        load_q_frags<UPW>(qr[k1], q2, (t + 1 + XD) * TQ, rows, C, heads, w, lane);
        load_d(dr[k], t + XD);
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            const int nt = w + 8 * i;
            const int row = row0 + QPF * nt + j;
            if (nt < FT && row < rows) {
                uint4* dp = reinterpret_cast<uint4*>(dxt + ((int64_t)row * FT + f) * C + cc * CH + 16 * kg);
                dp[0] = o[i][0];
                dp[1] = o[i][1];
            }
        }
        lds_barrier();                                           // one barrier per tile: U(t+1) complete, U(t) free
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // late ring loads: see time2_logits_kernel
}

// backward 2/2 with the q2 tile staged through LDS (see time2_logits_lds_kernel: the four fragment-layout q2 loads per wave
// were 32 of the 64 KB a tile pulled through the L1/TA at the 16 B/clk rate of that pattern).  q2 rows are loaded
// row-contiguous two tiles ahead, stored to a padded LDS image and read back as fragments; the small per-column operands
// (dl, a, dout) keep their direct loads, two tiles ahead.  Issue order per tile: q pieces (tile + 4), then the column
// operands (tile + 2): the counted waits below follow from it.
template <int FT, int UPW>
__global__ __launch_bounds__(512) void time2_dx_lds_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ wkT, int64_t ldw,
                                                           const float* __restrict__ attn2, const bf16_t* __restrict__ dl,
                                                           const bf16_t* __restrict__ dout, int64_t dobs,
                                                           bf16_t* __restrict__ dxt, int rows, int S, int heads, int spread) {
    constexpr int QPF = 16 / FT, NNT = FT > 8 ? FT / 8 : 1, KS = QPF > 2 ? QPF / 2 : 1, NHW = NHeads<UPW>::value;
    constexpr int NQP = 4, NX = NNT * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Owner own = owner_of_block(heads, spread);
    if (own.range < 0) return;
    const int C = heads * CH, cc = own.cc, QP = q_pitch(heads);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* sQb = smem + 2 * TQ * BWD_UROW;
    const int ntiles = (rows + TQ - 1) / TQ;
    const int t_begin = turn_begin(ntiles, own.range, own.nranges, 2), t_end = turn_begin(ntiles, own.range + 1, own.nranges, 2);
    const int niter = t_end - t_begin;
    if (niter <= 0) return;
    WFrag<UPW> wf;
    load_w_frags<UPW>(wf, wkT, ldw, cc, heads, w, lane);
    for (int e = tid; e < 2 * TQ * BWD_UROW / 16; e += 512)         // head rows >= heads stay zero
        reinterpret_cast<uint4*>(smem)[e] = make_uint4(0, 0, 0, 0);
    const int col = lane & 15, kg = lane >> 4, hfirst = (w * UPW) >> 1;
    const int j = col / FT, f = col % FT;
    const int hb = QPF >= 2 ? 8 * (kg >> 1) : 8 * (kg & 1);
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int qpr = C / 8, nqp = TQ * qpr;
    int qrow[NQP], qcol[NQP], qlds[NQP];
#pragma unroll
    for (int i = 0; i < NQP; ++i) {
        const int p = tid + 512 * i, pc = min(p, nqp - 1);
        qrow[i] = pc / qpr;
        qcol[i] = (pc - qrow[i] * qpr) * 8;
        qlds[i] = p < nqp ? qrow[i] * QP + qcol[i] * 2 : -1;
    }
    struct PreQ { bf16x8 q[NQP]; };
    struct DFrag { bf16x8 dl[NNT]; float a[NNT]; bf16x8 dout[NNT][2]; };
    auto issue_q = [&](PreQ& pr, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NQP; ++i) gload16_asm(pr.q[i], q2 + (int64_t)min(t * TQ + qrow[i], rows - 1) * C + qcol[i]);
    };
    auto pin_pq = [&](PreQ& pr) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NQP; ++i) pin(pr.q[i]);
    };
    auto stash_q = [&](const PreQ& pr, int stage) __attribute__((always_inline)) {
        char* sq = sQb + stage * (TQ * QP);
#pragma unroll
        for (int i = 0; i < NQP; ++i)
            if (qlds[i] >= 0) *reinterpret_cast<bf16x8*>(sq + qlds[i]) = pr.q[i];
    };
    auto read_q = [&](QFrag<NHW>& qf, int stage) __attribute__((always_inline)) {
        const char* sq = sQb + stage * (TQ * QP) + col * QP;
#pragma unroll
        for (int i = 0; i < NHW; ++i) {
            const int h = min(hfirst + i, heads - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf.v[i][ks] = *reinterpret_cast<const bf16x8*>(sq + (h * CH + 32 * ks + 8 * kg) * 2);
        }
    };
    auto load_d = [&](DFrag& x, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            const int nt = min(w + 8 * i, FT - 1);
            const int row = min(t * TQ + QPF * nt + j, rows - 1);
            const int b = row / S;
            gload16_asm(x.dl[i], dl + ((int64_t)row * FT + f) * MAXH + hb);
            gload4_asm(x.a[i], attn2 + ((int64_t)row * heads + cc) * FT + f);
            const bf16_t* dp = dout + b * dobs + (int64_t)(row - b * S) * C + cc * CH + 16 * kg;
            gload16_asm(x.dout[i][0], dp);
            gload16_asm(x.dout[i][1], dp + 8);
        }
    };
    auto pin_d = [&](DFrag& x) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NNT; ++i) { pin(x.dl[i]); pin(x.a[i]); pin(x.dout[i][0]); pin(x.dout[i][1]); }
    };
    PreQ pq[2];
    DFrag dr[2];
    // prologue: q2 tiles 0, 1 -> LDS; then, in the steady-state issue order of iterations -2 and -1: q(2) d(0) q(3) d(1)
    issue_q(pq[0], t_begin);
    issue_q(pq[1], t_begin + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin_pq(pq[0]); pin_pq(pq[1]);
    stash_q(pq[0], 0);
    stash_q(pq[1], 1);
    asm volatile("" ::: "memory");
    pin_pq(pq[0]); pin_pq(pq[1]);
    issue_q(pq[0], t_begin + 2);
    load_d(dr[0], t_begin);
    issue_q(pq[1], t_begin + 3);
    load_d(dr[1], t_begin + 1);
    lds_barrier();                                                // zero fill and q2 stages visible
    {
        QFrag<NHW> qf;
        read_q(qf, 0);
        compute_u_tile<UPW, true>(wf, qf, smem, 0, heads, w, lane);
    }
    lds_barrier();
    for (int itb = 0; itb < niter; itb += 2) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int it = itb + k;
        const int t = t_begin + it, row0 = t * TQ;
        const char* sU = smem + k * (TQ * BWD_UROW);
        // A: U^T of this tile (gathered h-major) and the next tile's q2 fragments -> registers
        TrFrag af[NNT][KS][4];
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            const int nt = min(w + 8 * i, FT - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int jp = QPF >= 2 ? (kg & 1) + 2 * ks : 0;
                const int sa = QPF * nt + jp;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    af[i][ks][mt].t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sU + bwd_off(sa, hb + tq, 4 * tp + mt)));
                    af[i][ks][mt].t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sU + bwd_off(sa, hb + 4 + tq, 4 * tp + mt)));
                }
            }
        }
        QFrag<NHW> qf;
        read_q(qf, k ^ 1);
        lds_barrier();                                           // every wave has what it needs of q2 stage k^1 ... and of
                                                                 // stage k (read one iteration ago): stage k may be rewritten
        // W / C: q2 tile it + 2 -> stage k, its registers re-issued for tile it + 4
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQP + 2 * NX) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pin_pq(pq[k]);
        stash_q(pq[k], k);
        asm volatile("" ::: "memory");
        pin_pq(pq[k]);
        issue_q(pq[k], t + 4);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQP + NX) : "memory");   // the column operands of this tile
        __builtin_amdgcn_sched_barrier(0);
        pin_d(dr[k]);
        // B: U of the next tile, dx~ of this one
        if (it + 1 < niter) compute_u_tile<UPW, true>(wf, qf, smem + (k ^ 1) * (TQ * BWD_UROW), 0, heads, w, lane);
        uint4 o[NNT][2];
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int jp = QPF >= 2 ? (kg & 1) + 2 * ks : 0;
                const bool keep = QPF >= 2 ? jp == j : kg < 2;
                bf16x8 bfr = dr[k].dl[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) bfr[e] = keep ? bfr[e] : (__bf16)0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks][mt].v, bfr, acc[mt], 0, 0, 0);
            }
            float dv[16], v[16];
            unpack8(*reinterpret_cast<const uint4*>(&dr[k].dout[i][0]), dv);
            unpack8(*reinterpret_cast<const uint4*>(&dr[k].dout[i][1]), dv + 8);
            const float av = dr[k].a[i];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * mt + r] = fmaf(av, dv[4 * mt + r], acc[mt][r]);
            o[i][0] = pack8(v);
            o[i][1] = pack8(v + 8);
            pin(o[i][0]); pin(o[i][1]);                          // the last use of this slot's operands is BEFORE its refill:
        }                                                        // see "leaving a slot" at pin()
        asm volatile("" ::: "memory");
        pin_d(dr[k]);
        load_d(dr[k], t + 2);
#pragma unroll
        for (int i = 0; i < NNT; ++i) {
            const int nt = w + 8 * i;
            const int row = row0 + QPF * nt + j;
            if (nt < FT && row < rows) {
                uint4* dp = reinterpret_cast<uint4*>(dxt + ((int64_t)row * FT + f) * C + cc * CH + 16 * kg);
                dp[0] = o[i][0];
                dp[1] = o[i][1];
            }
        }
        lds_barrier();                                           // U(t+1) and q2 stage k (tile it + 2) are complete
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // late ring loads: see time2_logits_kernel
}

int spread_mode() {
    static const int v = getenv("FOCUS_T2_SPREAD") ? atoi(getenv("FOCUS_T2_SPREAD")) : 0;
    return v;
}

size_t time2_lds_fwd(int heads) { return (size_t)2 * TQ * fwd_urow(heads); }
size_t time2_lds_bwd() { return (size_t)2 * TQ * BWD_UROW; }

}  // namespace

bool focus_traj_time2_ok(int F, int heads, int d, int dtype) {
    static const bool enabled = !(getenv("FOCUS_TIME2") && atoi(getenv("FOCUS_TIME2")) == 0);
    return enabled && dtype == FOCUS_BF16 && d == CH && heads >= 1 && heads <= MAXH && (F == 4 || F == 8 || F == 16) &&
           time2_lds_fwd(heads) <= 160 * 1024;
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
    const float scale = 1.f / sqrtf((float)d);
    const int upw = (2 * heads + 7) / 8;                          // weight units per wave (see WFrag)
    static const bool staged_on = !(getenv("FOCUS_T2_LDS") && atoi(getenv("FOCUS_T2_LDS")) == 0);
    const bool staged = staged_on && F <= 8 && time2_lds_staged(heads, F) <= 160 * 1024;
    const size_t lds = staged ? time2_lds_staged(heads, F) : time2_lds_fwd(heads);
#define TL(FT, UPW) do { \
        if (staged) { \
            static bool once_s = (hipFuncSetAttribute((const void*)time2_logits_lds_kernel<FT, UPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
            (void)once_s; \
            hipLaunchKernelGGL((time2_logits_lds_kernel<FT, UPW>), dim3(256), dim3(512), lds, s, (const bf16_t*)q2, (const bf16_t*)xt, (const bf16_t*)wkT, ldw, (float*)ws, (int)rows, heads, scale, spread_mode()); \
            break; } \
        static bool once = (hipFuncSetAttribute((const void*)time2_logits_kernel<FT, UPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
        (void)once; \
        hipLaunchKernelGGL((time2_logits_kernel<FT, UPW>), dim3(256), dim3(512), lds, s, (const bf16_t*)q2, (const bf16_t*)xt, (const bf16_t*)wkT, ldw, (float*)ws, (int)rows, heads, scale, spread_mode()); } while (0)
#define TLU(FT) do { if (upw == 1) TL(FT, 1); else if (upw == 2) TL(FT, 2); else if (upw == 3) TL(FT, 3); else TL(FT, 4); } while (0)
    if (F == 8) TLU(8); else if (F == 4) TLU(4); else TLU(16);
#undef TLU
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
    const int upw = (2 * heads + 7) / 8;
    static const bool staged_on = !(getenv("FOCUS_T2_LDS") && atoi(getenv("FOCUS_T2_LDS")) == 0);
    const size_t lds_staged = time2_lds_bwd() + (size_t)2 * TQ * q_pitch(heads);
    // (13..16 heads with F = 4 or 16: the staged variant would spill, and a spilled ring register is stored before its
    // inline-asm load lands -> those shapes keep the direct kernel and the staged one is not even instantiated: the
    // build's lint of hand-issued loads, focus_amd/build.py lint_hand_loads, refuses such a kernel)
    const bool staged = staged_on && lds_staged <= 160 * 1024 && !(upw == 4 && F != 8);
    const size_t lds = staged ? lds_staged : time2_lds_bwd();
#define TB(FT, UPW) do { \
        if constexpr (!((UPW) == 4 && (FT) != 8)) if (staged) { \
            static bool once_s = (hipFuncSetAttribute((const void*)time2_dx_lds_kernel<FT, UPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
            (void)once_s; \
            hipLaunchKernelGGL((time2_dx_lds_kernel<FT, UPW>), dim3(256), dim3(512), lds, s, (const bf16_t*)q2, (const bf16_t*)wkT, ldw, attn2, (const bf16_t*)dl, (const bf16_t*)dout, dout_bstride, (bf16_t*)dxt, (int)rows, S, heads, spread_mode()); \
            break; } \
        static bool once = (hipFuncSetAttribute((const void*)time2_dx_kernel<FT, UPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
        (void)once; \
        hipLaunchKernelGGL((time2_dx_kernel<FT, UPW>), dim3(256), dim3(512), lds, s, (const bf16_t*)q2, (const bf16_t*)wkT, ldw, attn2, (const bf16_t*)dl, (const bf16_t*)dout, dout_bstride, (bf16_t*)dxt, (int)rows, S, heads, spread_mode()); } while (0)
#define TBU(FT) do { if (upw == 1) TB(FT, 1); else if (upw == 2) TB(FT, 2); else if (upw == 3) TB(FT, 3); else TB(FT, 4); } while (0)
    if (F == 8) TBU(8); else if (F == 4) TBU(4); else TBU(16);
#undef TBU
#undef TB
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
