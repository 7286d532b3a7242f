// gemm_mfma_tn_ws.hip -- wave-specialised weight-gradient GEMM:  C[I,J] = alpha * sum_m P[m,I]^T . Q[m,J]
// (dW[N,K] = dY[M,N]^T . X[M,K] straight from the row-major activations; see gemm_mfma_tn.hip for the operand
// description and the transposed-LDS-read fragment gather, which this kernel shares).
//
// Why a second kernel: the 128x128 / 4-wave kernel asks L2 for 1 byte per 64 flop and keeps only one K-step of DMA
// in flight, and measured 290-430 TF/s.  Here a persistent 768-thread workgroup per CU runs
//   waves 0-7   consumers : 64x64 output tiles of a 256x128 (or 128x256) workgroup tile (85 flop per LDS byte),
//                           ds_read_b64_tr_b16 + v_mfma_f32_16x16x32_bf16 only;
//   waves 8-11  loaders   : global_load_lds_dwordx4 only, two K-steps ahead through a 3-stage 144 KiB ring,
//                           counted s_waitcnt vmcnt(12), one raw s_barrier per K-step.
// The reduction (rows m) is split over `splits` units per output tile; unit order is split-major so the units that
// run together read the same rows of P and Q (shared in the XCD's L2).  Each unit stores its fp32 partial tile
// to its slab aux[split][I][J]; tn_reduce (gemm_mfma_tn.hip) sums the slabs.  Atomic mode (no aux) adds into C.
#include "focus_common.h"
#include "gemm_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BKM = 64;                 // reduction rows per K-step
constexpr int SUB = BKM * 256;          // one [64 rows][128 columns] bf16 sub-tile: 16 KiB, 256-B rows
constexpr int NSTAGE = 3, NLOAD = 4;

union Frag { bf16x8 v; s16x4 t[2]; uint32_t u[4]; };

// same conflict-free swizzle as gemm_mfma_tn.hip (keyed for the transposed reads)
__device__ __forceinline__ int swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }
__device__ __forceinline__ int toff(int row, int col) {
    return row * 256 + ((((col >> 3)) ^ swz(row)) << 4) + (col & 4) * 2;
}
// k = tile rows r0..r0+7, m/n = tile column c0 + (lane & 15)
__device__ __forceinline__ bf16x8 col_frag16(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15;
    const int row = r0 + (i >> 2), col = c0 + 4 * (i & 3);
    Frag f;
    f.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + toff(row, col)));
    f.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + toff(row + 4, col)));
    return f.v;
}

struct Unit { int i0, j0, m_begin, nk, valid_last; };

template <int BI, int BJ>
__global__ __launch_bounds__(64 * (BI * BJ / 4096 + NLOAD)) void gemm_tn_ws_kernel(const focus_gemm_desc d, int tiles_i, int tiles_j,
                                                                                  int splits, int m_per_split, float* csum) {
    constexpr int NCONS = BI * BJ / 4096;
    constexpr int WJ = BJ / 64;
    constexpr int NSP = BI / 128, NSQ = BJ / 128, NSUBT = NSP + NSQ;       // sub-tiles per stage
    constexpr int STAGE = NSUBT * SUB;
    constexpr int PIECES = NSUBT * 16 / NLOAD;                            // 1 KiB DMA pieces per loader wave per K-step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mred = d.K;
    const bf16_t* Pm = static_cast<const bf16_t*>(d.A);     // P[m][i]
    const bf16_t* Qm = static_cast<const bf16_t*>(d.B);     // Q[m][j]
    const int64_t ldp = d.csA, ldq = d.rsB;

    // ---- unit schedule (identical for both roles): XCD-banded, split-major inside ----
    const int tiles = tiles_i * tiles_j, nunits = tiles * splits;
    const int G = gridDim.x, xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q = nunits >> 3, r = nunits & 7;
    const int band0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int band_n = q + (xcd < r ? 1 : 0);
    const int my_units = jwg < band_n ? (band_n - jwg + gx - 1) / gx : 0;
    if (my_units == 0) return;
    auto unit_of = [&](int i) __attribute__((always_inline)) {
        // an XCD owns a contiguous band of the split-major list: its resident workgroups share one slice of rows
        const int u = band0 + jwg + i * gx;
        const int split = u / tiles, tile = u % tiles;
        Unit t;
        t.i0 = (tile % tiles_i) * BI;
        t.j0 = (tile / tiles_i) * BJ;
        t.m_begin = split * m_per_split;
        const int len = min(Mred, t.m_begin + m_per_split) - t.m_begin;
        t.nk = (len + BKM - 1) / BKM;
        t.valid_last = len - (t.nk - 1) * BKM;
        return t;
    };

    if (w >= NCONS) {
        // =============================== loader waves ===============================
        const int L = w - NCONS;
        const int cpos = lane & 15, rin = lane >> 4;
        Unit cur = unit_of(0);
        int iu = 0, ikt = 0;
        auto issue = [&](int st) __attribute__((always_inline)) {
            char* sp = smem + st * STAGE;
            const int mb = cur.m_begin + ikt * BKM;
#pragma unroll
            for (int g = 0; g < PIECES; ++g) {
                // piece g*NLOAD + L: the sub-tile index g>>2 is a compile-time constant per unrolled iteration
                const int sub = g >> 2, pi = g * NLOAD + L, row = (pi & 15) * 4 + rin;
                const int m = min(mb + row, Mred - 1);
                const int ch = cpos ^ swz(row);
                const bf16_t* src;
                if (sub < NSP) src = Pm + (int64_t)m * ldp + min(cur.i0 + sub * 128 + ch * 8, d.M - 8);
                else src = Qm + (int64_t)m * ldq + min(cur.j0 + (sub - NSP) * 128 + ch * 8, d.N - 8);
                __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(sp + pi * 1024), 16, 0, 0);
            }
            if (++ikt == cur.nk) { ikt = 0; if (++iu < my_units) cur = unit_of(iu); }
        };
        int total = 0;
        for (int i = 0; i < my_units; ++i) total += unit_of(i).nk;
        issue(0);
        if (total > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                      // step 0 is in LDS
        int st2 = 2;
        for (int t = 0; t < total; ++t) {
            if (t + 2 < total) {
                issue(st2);
                st2 = st2 == 2 ? 0 : st2 + 1;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");     // step t+1 landed, t+2 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                  // end of K-step t
        }
        return;
    }

    // =============================== consumer waves ===============================
    const int wi = w / WJ, wj = w % WJ;
    const int fr = lane & 15, fq = lane >> 4;
    const int poff = (wi >> 1) * SUB, pcol = (wi & 1) * 64;
    const int qoff = (NSP + (wj >> 1)) * SUB, qcol = (wj & 1) * 64;
    f32x4 acc[4][4];
    // Optional column sums of P (bias gradient: db[i] = sum_m dY[m,i]) on the matrix pipe: D = ones^T . P gives
    // sum_k P[k][i] in every row.  The work is dealt over the units of a row of tiles (unit tile_j takes the K-steps
    // kt = tile_j mod tiles_j) and over the WJ waves that share the same P fragments, so it costs each unit about
    // 1/(4*tiles_j) extra MFMAs; partial sums go to csum[split*tiles_j + tile_j][M] (summed by tn_reduce).
    constexpr int NA = 4 / WJ;
    f32x4 accb[NA];
    Frag ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3F803F80u;
    // valid < BKM only on the last K-step of a unit that ends at the end of the reduction: rows past it were DMA'd
    // from a clamped address, so their P elements are zeroed in the fragment (element e <-> row ks*32 + 8*fq + e)
    auto compute = [&](const char* stage, auto masked, int valid, bool cs) __attribute__((always_inline)) {
        const char* sp = stage + poff;
        const char* sq = stage + qoff;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fp[4], fqv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fp[a] = col_frag16(sp, ks * 32 + 8 * fq, pcol + a * 16, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) fqv[b] = col_frag16(sq, ks * 32 + 8 * fq, qcol + b * 16, lane);
            if constexpr (decltype(masked)::value) {
                const int left = valid - (ks * 32 + 8 * fq);            // elements e < left are real
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    Frag f; f.v = fp[a];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const uint32_t keep = (2 * e2 < left ? 0x0000ffffu : 0u) | (2 * e2 + 1 < left ? 0xffff0000u : 0u);
                        f.u[e2] &= keep;
                    }
                    fp[a] = f.v;
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqv[b], fp[a], acc[a][b], 0, 0, 0);
            if (cs) {                                   // wave-uniform; touches accb only (acc stays branch-free)
#pragma unroll
                for (int n = 0; n < NA; ++n) {
                    bf16x8 sel = fp[n];
#pragma unroll
                    for (int j = 1; j < WJ; ++j)
                        if (wj == j) sel = fp[j * NA + n];
                    accb[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, sel, accb[n], 0, 0, 0);
                }
            }
        }
    };

    __builtin_amdgcn_s_barrier();                          // step 0 is in LDS
    int st = 0;
    for (int cu = 0; cu < my_units; ++cu) {
        const Unit cur = unit_of(cu);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // straight-line structure (no branch between compute variants inside the loop: a branch there makes the
        // compiler keep two copies of the accumulators): nk-1 plain steps, then the last step always masked
        const int tj = cur.j0 / BJ;
        int cs_in = tj;                                    // steps until this unit's next column-sum K-step
#pragma unroll
        for (int n = 0; n < NA; ++n) accb[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt + 1 < cur.nk; ++kt) {
            compute(smem + st * STAGE, std::false_type{}, BKM, csum != nullptr && cs_in == 0);
            cs_in = cs_in == 0 ? tiles_j - 1 : cs_in - 1;
            st = st == 2 ? 0 : st + 1;
            __builtin_amdgcn_s_barrier();                  // end of this K-step
        }
        compute(smem + st * STAGE, std::true_type{}, cur.valid_last, csum != nullptr && cs_in == 0);
        st = st == 2 ? 0 : st + 1;
        __builtin_amdgcn_s_barrier();
        // acc[a][b][r4] = D[j = j0 + wj*64 + b*16 + fq*4 + r4][i = i0 + wi*64 + a*16 + fr]; stores straight from the
        // registers (no LDS), so the loaders keep filling the ring for the next unit meanwhile
        const int split = cur.m_begin / m_per_split;
        if (csum && fq == 0) {
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int gi = cur.i0 + wi * 64 + (wj * NA + n) * 16 + fr;
                if (gi < d.M) csum[(int64_t)(split * tiles_j + tj) * d.M + gi] = d.alpha * accb[n][0];
            }
        }
        float* C = d.aux ? static_cast<float*>(d.aux) + (int64_t)split * d.M * d.N : static_cast<float*>(d.C);
        const int64_t ldc = d.aux ? d.N : d.rsC;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int gi = cur.i0 + wi * 64 + a * 16 + fr;
            if (gi >= d.M) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int gj = cur.j0 + wj * 64 + b * 16 + fq * 4;
                if (d.aux) {
                    if (gj < d.N)      // N % 8 == 0: the 4 columns are in range together
                        *reinterpret_cast<float4*>(C + (int64_t)gi * ldc + gj) =
                            make_float4(d.alpha * acc[a][b][0], d.alpha * acc[a][b][1], d.alpha * acc[a][b][2], d.alpha * acc[a][b][3]);
                } else {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        if (gj + r4 < d.N) atomicAdd(C + (int64_t)gi * ldc + (gj + r4), d.alpha * acc[a][b][r4]);
                }
            }
        }
    }
}

template <int BI, int BJ>
int launch_tn_ws(const focus_gemm_desc& d, const focus_tn_plan& pl, float* csum, hipStream_t s) {
    const size_t lds = (size_t)NSTAGE * (BI / 128 + BJ / 128) * SUB;
    auto k = gemm_tn_ws_kernel<BI, BJ>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    const int nunits = pl.tiles_i * pl.tiles_j * pl.splits;
    hipLaunchKernelGGL(k, dim3(std::min(nunits, 256)), dim3(64 * (BI * BJ / 4096 + NLOAD)), lds, s, d, pl.tiles_i, pl.tiles_j,
                       pl.splits, pl.m_per_split, csum);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

// Plan for the wave-specialised kernel; kind == 0 means "not worth it" (small outputs / short reductions).
focus_tn_plan focus_gemm_tn_ws_plan(int M, int N, int K) {
    static const bool enabled = !(getenv("FOCUS_GEMM_TN_WS") && atoi(getenv("FOCUS_GEMM_TN_WS")) == 0);
    static const int rounds_env = getenv("FOCUS_GEMM_TN_ROUNDS") ? atoi(getenv("FOCUS_GEMM_TN_ROUNDS")) : 0;
    focus_tn_plan p = {};
    if (!enabled || M < 128 || N < 128 || K < 2048) return p;
    // the longer output side takes the 256 edge (fewer partial tiles hanging over the matrix edge) -- unless the output is
    // narrower than one 256 tile and the other orientation pads strictly less: [dk|dv]^T x of STEVE is 384 x 192, 4 tiles of 256 x 128 cover 512 x 256 (56 % useful
    // MFMA work), 3 tiles of 128 x 256 cover 384 x 256 (75 %)
    const int64_t pad1 = (int64_t)((M + 255) / 256 * 256) * ((N + 127) / 128 * 128);
    const int64_t pad2 = (int64_t)((M + 127) / 128 * 128) * ((N + 255) / 256 * 256);
    bool wide_i = (M >= N || N % 256 != 0) && M >= 256;
    static const bool pad_rule = !(getenv("FOCUS_GEMM_TN_PAD_RULE") && atoi(getenv("FOCUS_GEMM_TN_PAD_RULE")) == 0);
    // (only for N < 256: at 384 x 768, ORViT's patch_to_d gradient, the flip measured 0.6 % of the step slower)
    if (wide_i && pad_rule && pad2 < pad1 && N < 256) wide_i = false;
    if (!wide_i && N < 256 && !(pad2 < pad1)) return p;
    const int bi = wide_i ? 256 : 128, bj = wide_i ? 128 : 256;
    p.kind = wide_i ? 1 : 2;
    p.tiles_i = (M + bi - 1) / bi;
    p.tiles_j = (N + bj - 1) / bj;
    const int tiles = p.tiles_i * p.tiles_j;
    const int max_splits = std::max(1, K / (4 * BKM));           // at least 4 K-steps per unit
    // one round of 256 persistent workgroups: more splits only add slab traffic (measured: 2 and 3 rounds are
    // 5-30 % slower on every hot shape); FOCUS_GEMM_TN_ROUNDS overrides for tuning
    const int rounds = rounds_env > 0 ? rounds_env : 1;
    const int best_s = std::min(max_splits, std::max(1, 256 * rounds / tiles));
    p.splits = best_s;
    p.m_per_split = ((K + p.splits - 1) / p.splits + BKM - 1) / BKM * BKM;
    p.splits = (K + p.m_per_split - 1) / p.m_per_split;
    return p;
}

// csum: NULL, or [splits * tiles_j][M] fp32 partial column sums of the A operand (see the kernel)
int focus_gemm_mfma_tn_ws(const focus_gemm_desc& d, const focus_tn_plan& pl, float* csum, hipStream_t s) {
    if (pl.kind == 1) return launch_tn_ws<256, 128>(d, pl, csum, s);
    if (pl.kind == 2) return launch_tn_ws<128, 256>(d, pl, csum, s);
    return FOCUS_ERR_SHAPE;
}
