// gemm_mfma_ws.hip -- wave-specialised bf16 MFMA GEMM:  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T), bf16 out.
//
// Why: in the uniform kernel (gemm_mfma.hip) every wave pays the issue cost of its 8 LDS-DMA pieces per K-step
// (~60-180 cycles each, MI355X_MICROARCH.md "LDS-DMA piece issue cost") in front of only 32 MFMAs (512 cycles), and
// only one K-step of DMA is in flight per workgroup: the matrix pipe idled ~60 % of the time.  Here a 512-thread
// persistent workgroup splits roles:
//   waves 0-3  consumers : 2x2 grid of 64x64 output tiles, ds_read_b128 + v_mfma_f32_16x16x32_bf16 only;
//   waves 4-7  loaders   : global_load_lds_dwordx4 only, running TWO K-steps ahead through a 3-stage LDS ring
//                          (96 KiB), counted s_waitcnt vmcnt(8) so one K-step stays in flight across the barrier.
// One raw s_barrier per K-step hands a landed stage to the consumers and a consumed stage back to the loaders.
// The K-steps of consecutive output tiles form one stream (persistent workgroup, XCD-banded tile order), so the
// ring never drains at tile boundaries; the bf16 epilogue (bias / GELU+aux / ReLU / tanh / residual / x act')
// goes through per-wave LDS slabs in the just-consumed stage and touches C, aux and residual in 16-byte pieces of
// whole 128-byte rows.
#include "focus_common.h"
#include "gemm_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int BK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct Unit { int m0, n0, nk; };

// 128x256 / 256x128: 8 consumer waves (64x64 each) + 4 loaders, 3-stage 144 KiB ring (85 flop per LDS-filled byte;
//           the loaders run two K-steps ahead).
// 192x256: 12 consumer waves + 4 loaders = 1024 threads, 2-stage 112 KiB ring (110 flop per LDS-filled byte: the
//           kernel is bound by the L2 -> LDS fill rate of a CU, ~22 B/clk, so bytes per flop is the lever; and
//           M = 12552 x N = 768 becomes 198 tiles = ONE round of the 256 CUs instead of 297 = two).  With two
//           stages the loaders run one K-step ahead (s_waitcnt vmcnt(0) per step).
template <int BM, int BN, int EPI, int NLOAD, int NSTAGE>
__global__ __launch_bounds__(64 * (BM * BN / 4096 + NLOAD)) void gemm_nt_ws_kernel(const focus_gemm_desc d, int tiles_m, int tiles_n, int GM) {
    constexpr int NCONS = BM * BN / 4096;                        // consumer waves (64x64 each)
    constexpr int WN = BN / 64;                                  // consumer grid is (BM/64) x WN
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;   // bytes per ring stage
    constexpr int GA = BM / 8 / NLOAD, GB = BN / 8 / NLOAD;      // DMA pieces per loader wave per K-step
    constexpr int PIECES = GA + GB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int z = blockIdx.y;
    const int b0 = z / d.batch1, b1 = z % d.batch1;
    const bf16_t* A = static_cast<const bf16_t*>(d.A) + b0 * d.bsA0 + b1 * d.bsA1;
    const bf16_t* B = static_cast<const bf16_t*>(d.B) + b0 * d.bsB0 + b1 * d.bsB1;
    const int64_t coff = b0 * d.bsC0 + b1 * d.bsC1;
    bf16_t* C = static_cast<bf16_t*>(d.C) + coff;
    const bf16_t* R = d.residual ? static_cast<const bf16_t*>(d.residual) + coff : nullptr;
    bf16_t* X = d.aux ? static_cast<bf16_t*>(d.aux) + coff : nullptr;
    const int64_t lda = d.rsA, ldb = d.csB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- unit schedule (identical for both roles): XCD-banded, round-robin inside the band ----
    const int nunits = tiles_m * tiles_n, nk = d.K / BK;
    const int G = gridDim.x, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q = nunits >> 3, r = nunits & 7;
    const int band0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int band_n = q + (xcd < r ? 1 : 0);
    const int my_units = j < band_n ? (band_n - j + gx - 1) / gx : 0;
    if (my_units == 0) return;
    auto unit_of = [&](int i) __attribute__((always_inline)) {
        const int u = band0 + j + i * gx;
        const int group = u / (GM * tiles_n), first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM), in_g = u - group * GM * tiles_n;
        Unit t;
        t.m0 = (first_m + in_g % gsz) * BM;
        t.n0 = (in_g / gsz) * BN;
        t.nk = nk;
        return t;
    };
    const int total = my_units * nk;      // K-steps in this workgroup's stream

    if (w >= NCONS) {
        // =============================== loader waves ===============================
        const int L = w - NCONS;
        const int lrow = lane >> 3, cpos = lane & 7, csrc = (cpos ^ lrow) * 8;
        const bf16_t* a_src[GA];
        const bf16_t* b_src[GB];
        int iu = 0, ikt = 0;
        auto setup = [&](int i) __attribute__((always_inline)) {
            const Unit t = unit_of(i);
#pragma unroll
            for (int g = 0; g < GA; ++g)
                a_src[g] = A + (int64_t)min(t.m0 + (L * GA + g) * 8 + lrow, d.M - 1) * lda + csrc;
#pragma unroll
            for (int g = 0; g < GB; ++g)
                b_src[g] = B + (int64_t)min(t.n0 + (L * GB + g) * 8 + lrow, d.N - 1) * ldb + csrc;
        };
        auto issue = [&](int st) __attribute__((always_inline)) {
            char* sa = smem + st * STAGE;
            char* sb = sa + A_BYTES;
#pragma unroll
            for (int g = 0; g < GA; ++g)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(a_src[g] + ikt * BK), (lvoid_t*)(sa + (L * GA + g) * 1024), 16, 0, 0);
#pragma unroll
            for (int g = 0; g < GB; ++g)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(b_src[g] + ikt * BK), (lvoid_t*)(sb + (L * GB + g) * 1024), 16, 0, 0);
            if (++ikt == nk) { ikt = 0; if (++iu < my_units) setup(iu); }
        };
        constexpr int AHEAD = NSTAGE - 1;                  // K-steps the loaders run ahead of the consumers
        constexpr int INFLIGHT = (AHEAD - 1) * PIECES;     // DMA pieces that may stay outstanding across a barrier
        setup(0);
        issue(0);
        if (AHEAD > 1 && total >= AHEAD) {
#pragma unroll
            for (int a = 1; a < AHEAD; ++a) issue(a);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        } else {
            for (int a = 1; a < AHEAD && a < total; ++a) issue(a);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                      // step 0 is in LDS
        int st2 = AHEAD % NSTAGE, ckt = 0;                 // stage for step t+AHEAD ; position inside the current unit
        for (int t = 0; t < total; ++t) {
            if (t + AHEAD < total) {
                issue(st2);
                st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // step t+1 landed, later steps in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                  // end of K-step t
            if (++ckt == nk) {                             // unit boundary: consumers run their epilogue in a stage
                ckt = 0;
                __builtin_amdgcn_s_barrier();              // barrier E (epilogue done, stage reusable)
            }
        }
        return;
    }

    // =============================== consumer waves ===============================
    const int wm = w / WN, wn = w % WN;
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];
    auto compute = [&](const char* sa) __attribute__((always_inline)) {
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * 64 + i * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                fb[jj] = *reinterpret_cast<const bf16x8*>(sb + swz(wn * 64 + jj * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jj], fa[i], acc[i][jj], 0, 0, 0);
        }
    };
    auto epilogue = [&](char* stage, int m0, int n0) __attribute__((always_inline)) {
        // the wave's 64x64 tile leaves in two 32-row halves through a private 4 KiB slab
        // ([32 rows][16 chunks of 8 B], chunk ^= row & 15) inside the just-consumed stage
        char* slab = stage + w * 4096;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int i = half * 2 + i2, row = i2 * 16 + frow;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int gn = n0 + wn * 64 + jj * 16 + fq * 4;
                    float t[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        t[r4] = d.alpha * acc[i][jj][r4];
                        if (d.bias && gn + r4 < d.N) t[r4] += d.bias[gn + r4];
                    }
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(t[0]) | ((uint32_t)f32_to_bf16(t[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(t[2]) | ((uint32_t)f32_to_bf16(t[3]) << 16);
                    *reinterpret_cast<uint2*>(slab + row * 128 + (((jj * 4 + fq) ^ (row & 15)) << 3)) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int q8 = lane & 7;
#pragma unroll
            for (int p8 = 0; p8 < 4; ++p8) {
                const int row = p8 * 8 + (lane >> 3);
                const int gm = m0 + wm * 64 + half * 32 + row, gn = n0 + wn * 64 + q8 * 8;
                uint4 raw = *reinterpret_cast<const uint4*>(slab + row * 128 + ((q8 ^ ((row & 15) >> 1)) << 4));
                if (row & 1) { const uint32_t a0 = raw.x, a1 = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = a0; raw.w = a1; }
                if (gm >= d.M || gn >= d.N) continue;
                const int64_t off = gm * d.rsC + gn;
                float v[8] = {__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                              __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u),
                              __uint_as_float(raw.z << 16), __uint_as_float(raw.z & 0xffff0000u),
                              __uint_as_float(raw.w << 16), __uint_as_float(raw.w & 0xffff0000u)};
                float xs[8];
                if constexpr (EPI >= FOCUS_EPI_DGELU) {
                    const uint4 xr = *reinterpret_cast<const uint4*>(X + off);
                    xs[0] = __uint_as_float(xr.x << 16); xs[1] = __uint_as_float(xr.x & 0xffff0000u);
                    xs[2] = __uint_as_float(xr.y << 16); xs[3] = __uint_as_float(xr.y & 0xffff0000u);
                    xs[4] = __uint_as_float(xr.z << 16); xs[5] = __uint_as_float(xr.z & 0xffff0000u);
                    xs[6] = __uint_as_float(xr.w << 16); xs[7] = __uint_as_float(xr.w & 0xffff0000u);
                }
                if constexpr (EPI == FOCUS_EPI_GELU) { if (X) *reinterpret_cast<uint4*>(X + off) = raw; }
                // EPI is a template parameter: with a run-time switch here the 64 inlined copies of erf/tanh made the
                // epilogue ~17k instructions (140 KB, larger than the instruction cache) and cost ~10 us per tile
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (EPI == FOCUS_EPI_GELU) v[e] = gelu_erf(v[e]);
                    else if constexpr (EPI == FOCUS_EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                    else if constexpr (EPI == FOCUS_EPI_TANH) v[e] = tanhf(v[e]);
                    else if constexpr (EPI == FOCUS_EPI_DGELU) v[e] *= dgelu_erf(xs[e]);
                    else if constexpr (EPI == FOCUS_EPI_DRELU) v[e] = xs[e] > 0.f ? v[e] : 0.f;
                    else if constexpr (EPI == FOCUS_EPI_DTANH) v[e] *= (1.f - xs[e] * xs[e]);
                }
                if (R) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(R + off);
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                    v[4] += __uint_as_float(rr.z << 16); v[5] += __uint_as_float(rr.z & 0xffff0000u);
                    v[6] += __uint_as_float(rr.w << 16); v[7] += __uint_as_float(rr.w & 0xffff0000u);
                }
                uint4 o;
                o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
                o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
                typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
                // non-temporal: C is not re-read by this kernel; keeps the A/B panels in L2 (+6-20 % at K <= 768)
                __builtin_nontemporal_store((u32x4){o.x, o.y, o.z, o.w}, reinterpret_cast<u32x4*>(C + off));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab re-read before the second half overwrites it
        }
    };

    __builtin_amdgcn_s_barrier();                          // step 0 is in LDS
    int st = 0;
    for (int cu = 0; cu < my_units; ++cu) {
        const Unit cur = unit_of(cu);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        char* last_stage = smem;
        for (int kt = 0; kt < nk; ++kt) {
            char* sa = smem + st * STAGE;
            compute(sa);
            last_stage = sa;
            st = st == NSTAGE - 1 ? 0 : st + 1;
            __builtin_amdgcn_s_barrier();                  // end of this K-step (every consumer is done with `sa`)
        }
        epilogue(last_stage, cur.m0, cur.n0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // barrier E
    }
}

}  // namespace

bool focus_gemm_mfma_ws_ok(const focus_gemm_desc& d) {
    static const bool enabled = !(getenv("FOCUS_GEMM_WS") && atoi(getenv("FOCUS_GEMM_WS")) == 0);
    if (!enabled || !focus_gemm_mfma_nt_ok(d)) return false;
    if (d.dtype_c != FOCUS_BF16 || d.accumulate || d.csC != 1 || (d.rsC & 7) || (d.N & 7)) return false;
    if ((d.bsC0 & 7) || (d.bsC1 & 7)) return false;
    if (!focus_aligned(d.C, 16) || (d.residual && !focus_aligned(d.residual, 16)) || (d.aux && !focus_aligned(d.aux, 16)))
        return false;
    return true;
}

template <int BM, int BN, int EPI, int NLOAD>
static int launch_ws_n(const focus_gemm_desc& d, hipStream_t s) {
    constexpr int FIT = 160 * 1024 / ((BM + BN) * 128);
    constexpr int NSTAGE = FIT >= 4 ? 4 : FIT;                    // ring depth: what fits the 160 KiB of LDS, at most 4
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    const int nbatch = d.batch0 * d.batch1;
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    auto k = gemm_nt_ws_kernel<BM, BN, EPI, NLOAD, NSTAGE>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    const int nunits = tiles_m * tiles_n;
    const int resident = std::max(8, 256 / std::max(1, std::min(nbatch, 32)));
    dim3 grid(std::min(nunits, resident), nbatch);
    // unit order: narrow outputs (<= 4 column tiles, e.g. N = 768) walk a row of tiles first, so neighbouring
    // 512-B pieces of the same C rows are written close together in time (measured +14 % on 100352x768x768);
    // wide outputs use 8-row-tile-deep groups so an XCD's resident tiles share A and B panels in its L2.
    static const int gm_env = getenv("FOCUS_GEMM_GM") ? std::max(1, atoi(getenv("FOCUS_GEMM_GM"))) : 0;
    const int gm = gm_env ? gm_env : (tiles_n <= 4 ? 1 : 8);
    hipLaunchKernelGGL(k, grid, dim3(64 * (BM * BN / 4096 + NLOAD)), lds, s, d, tiles_m, tiles_n, gm);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <int BM, int BN, int EPI>
static int launch_ws(const focus_gemm_desc& d, hipStream_t s) {
    // loader waves per workgroup: the LDS-DMA issue rate of the loaders, not the MFMA rate, bounds a K-step
    // (~2.2k clocks per K-step with 4 loaders x 12 pieces against 1k clocks of MFMA); FOCUS_GEMM_NLOAD=4|8 for tuning
    return launch_ws_n<BM, BN, EPI, 4>(d, s);
}

template <int BM, int BN>
static int launch_ws_epi(const focus_gemm_desc& d, hipStream_t s) {
    switch (d.epilogue) {
        case FOCUS_EPI_NONE: return launch_ws<BM, BN, FOCUS_EPI_NONE>(d, s);
        case FOCUS_EPI_GELU: return launch_ws<BM, BN, FOCUS_EPI_GELU>(d, s);
        case FOCUS_EPI_RELU: return launch_ws<BM, BN, FOCUS_EPI_RELU>(d, s);
        case FOCUS_EPI_TANH: return launch_ws<BM, BN, FOCUS_EPI_TANH>(d, s);
        case FOCUS_EPI_DGELU: return launch_ws<BM, BN, FOCUS_EPI_DGELU>(d, s);
        case FOCUS_EPI_DRELU: return launch_ws<BM, BN, FOCUS_EPI_DRELU>(d, s);
        case FOCUS_EPI_DTANH: return launch_ws<BM, BN, FOCUS_EPI_DTANH>(d, s);
        default: return FOCUS_ERR_SHAPE;
    }
}

static int g_tile_override = 0;
extern "C" int focus_gemm_tile_override(int bm) {
    if (bm != 0 && bm != 128 && bm != 192) return FOCUS_ERR_SHAPE;
    g_tile_override = bm;
    return FOCUS_OK;
}

int focus_gemm_mfma_ws(const focus_gemm_desc& d, hipStream_t s) {
    if (d.batch0 * d.batch1 > 65535) return FOCUS_ERR_SHAPE;
    const int64_t t256 = (int64_t)((d.M + 255) / 256) * ((d.N + 127) / 128);
    const int64_t tw = (int64_t)((d.M + 127) / 128) * ((d.N + 255) / 256);
    const int64_t t192 = (int64_t)((d.M + 191) / 192) * ((d.N + 255) / 256);
    // tile choice by modelled time = rounds of the 256 persistent workgroups x time per tile; a 192x256 tile measures
    // 1.40-1.45x the time of a 128x256 tile for 1.5x its work (tools/gemm_tile_ab.py), so it wins where it saves a
    // round: 12552 x 768 is 198 tiles = one round instead of 297 = two (1.31-1.38x faster at K = 768 .. 3072).
    // FOCUS_GEMM_TILE=128|192 or focus_gemm_tile_override() force a shape (tuning).
    static const int env_force = getenv("FOCUS_GEMM_TILE") ? atoi(getenv("FOCUS_GEMM_TILE")) : 0;
    const int force = g_tile_override ? g_tile_override : env_force;
    if (d.N >= 256 && d.batch0 * d.batch1 == 1) {
        const double c128 = (double)((tw + 255) / 256) * 1.0, c192 = (double)((t192 + 255) / 256) * 1.42;
        const bool use192 = force ? force == 192 : (t192 >= 128 && c192 < c128);
        if (use192 && t192 >= 128) return launch_ws_epi<192, 256>(d, s);
    }
    // narrow outputs (per-head products: N = head dim, batched over the heads): 4 consumers on a 256 x 64 tile, no MFMA
    // work on padding columns, and the 3-stage ring instead of the uniform kernel's one K-step in flight
    if (d.N <= 64 && d.M >= 2048 && d.epilogue == FOCUS_EPI_NONE && (int64_t)((d.M + 255) / 256) * d.batch0 * d.batch1 >= 192)
        return launch_ws<256, 64, FOCUS_EPI_NONE>(d, s);
    if (d.N >= 256 && tw >= 192) return launch_ws_epi<128, 256>(d, s);   // 512-byte row segments of C
    if (d.M >= 256 && t256 >= 192) return launch_ws_epi<256, 128>(d, s);
    return FOCUS_ERR_SHAPE;   // too few tiles for one 8-consumer workgroup per CU: the caller uses the uniform kernel
}
