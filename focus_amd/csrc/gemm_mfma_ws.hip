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

// Consumer grid WM x WN waves, each wave a (16 MI) x 64 output tile (MI fragments of 16 rows, 4 of 16 columns):
//   128x256 / 256x128 (MI 4): 8 consumers + 4 loaders, 3-stage 144 KiB ring (85 flop per LDS-filled byte; the loaders
//           run two K-steps ahead).
//   160x256 (MI 5), 192x256 (MI 6): 8 consumers + 4 loaders.  [r3] The tile height is a free parameter in steps of 32
//           rows, chosen per shape so that the tiles fill whole rounds of the 256 CUs: M = 12552 x N = 768 is 237 tiles
//           of 160 rows = ONE round at 92 % (198 tiles of 192 rows: 77 %; 297 of 128: two rounds at 58 %), N = 2304 three
//           rounds of 160 instead of four of 128.  160x256 keeps a 3-stage ring (156 KiB).
//   192x256 with 12 consumers of 64x64 (MI 4, WM 3): the round-2 form, kept for A/B (FOCUS_GEMM_TILE=193).
// BFP8: the B operand (the weights) is stored as OCP e4m3 codes, 64 bytes per row and K-step: half the B bytes through
//           the LDS ring; consumers read 8 codes per fragment (ds_read_b64) and widen them to bf16 with
//           v_cvt_scalef32_pk_bf16_fp8 (exact: every e4m3 value is a bf16 value); the per-tensor scale multiplies the
//           fp32 accumulator in the epilogue.  "fp8 weights, bf16 activations" (BASELINE configs[4]).
template <int WM, int WN, int MI, int EPI, int NLOAD, int NSTAGE, bool BFP8>
__global__ __launch_bounds__(64 * (WM * WN + NLOAD)) void gemm_nt_ws_kernel(const focus_gemm_desc d, int tiles_m, int tiles_n, int GM) {
    constexpr int BM = WM * MI * 16, BN = WN * 64;
    constexpr int NCONS = WM * WN;                               // consumer waves
    constexpr int BROW = BFP8 ? 64 : 128;                        // bytes of one B row per K-step
    constexpr int A_BYTES = BM * 128, STAGE = BM * 128 + BN * BROW;   // bytes per ring stage
    constexpr int GA = BM / 8 / NLOAD, GB = BN * BROW / 1024 / NLOAD; // DMA pieces (1 KiB) per loader wave per K-step
    static_assert(GA * 8 * NLOAD == BM && GB * 1024 * NLOAD == BN * BROW, "pieces must divide among the loader waves");
    constexpr int PIECES = GA + GB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int z = blockIdx.y;
    const int b0 = z / d.batch1, b1 = z % d.batch1;
    const bf16_t* A = static_cast<const bf16_t*>(d.A) + b0 * d.bsA0 + b1 * d.bsA1;
    const char* B = static_cast<const char*>(d.B) + (b0 * d.bsB0 + b1 * d.bsB1) * (BFP8 ? 1 : 2);
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
        // fp8 B: a 1 KiB piece is 16 rows of 64 bytes; lane -> (row lane>>2, 16-byte chunk lane&3), the chunk a row keeps at
        // position p is chunk p ^ ((row >> 2) & 3): the consumers' ds_read_b64 (16 rows x 2 halves per 32-lane group) then
        // touch every bank once
        const int brow8 = lane >> 2, bsrc8 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
        const bf16_t* a_src[GA];
        const char* b_src[GB];
        int iu = 0, ikt = 0;
        auto setup = [&](int i) __attribute__((always_inline)) {
            const Unit t = unit_of(i);
#pragma unroll
            for (int g = 0; g < GA; ++g)
                a_src[g] = A + (int64_t)min(t.m0 + (L * GA + g) * 8 + lrow, d.M - 1) * lda + csrc;
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                if constexpr (BFP8) b_src[g] = B + (int64_t)min(t.n0 + (L * GB + g) * 16 + brow8, d.N - 1) * ldb + bsrc8;
                else b_src[g] = B + ((int64_t)min(t.n0 + (L * GB + g) * 8 + lrow, d.N - 1) * ldb + csrc) * 2;
            }
        };
        auto issue = [&](int st) __attribute__((always_inline)) {
            char* sa = smem + st * STAGE;
            char* sb = sa + A_BYTES;
#pragma unroll
            for (int g = 0; g < GA; ++g)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(a_src[g] + ikt * BK), (lvoid_t*)(sa + (L * GA + g) * 1024), 16, 0, 0);
#pragma unroll
            for (int g = 0; g < GB; ++g)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(b_src[g] + ikt * (BFP8 ? BK : 2 * BK)), (lvoid_t*)(sb + (L * GB + g) * 1024), 16, 0, 0);
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
    const float alpha = BFP8 ? d.alpha * *d.b_scale : d.alpha;
    f32x4 acc[MI][4];
    auto compute = [&](const char* sa) __attribute__((always_inline)) {
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[MI], fb[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if constexpr (BFP8) {
                    // 8 e4m3 codes k = 32 ks + 8 fq .. +7 of row n: 16-byte chunk 2 ks + (fq >> 1), half fq & 1
                    const int row = wn * 64 + jj * 16 + frow;
                    const uint2 raw = *reinterpret_cast<const uint2*>(sb + row * 64 + (((2 * ks + (fq >> 1)) ^ ((row >> 2) & 3)) << 4) + (fq & 1) * 8);
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    union { bf16x2 p[4]; bf16x8 v; } u;
                    u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, false);
                    u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, true);
                    u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, false);
                    u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, true);
                    fb[jj] = u.v;
                } else {
                    fb[jj] = *reinterpret_cast<const bf16x8*>(sb + swz(wn * 64 + jj * 16 + frow, ks * 4 + fq));
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * (16 * MI) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < MI; ++i)
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
        for (int half = 0; half < (MI + 1) / 2; ++half) {
            constexpr int LASTN = (MI & 1) ? 1 : 2;                 // fragment rows in the last pass (odd MI: one)
            const int nfr = (half == (MI + 1) / 2 - 1) ? LASTN : 2;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                if (i2 >= nfr) break;
                const int i = half * 2 + i2, row = i2 * 16 + frow;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int gn = n0 + wn * 64 + jj * 16 + fq * 4;
                    float t[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        t[r4] = alpha * acc[i][jj][r4];
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
                if (p8 >= 2 * nfr) break;
                const int row = p8 * 8 + (lane >> 3);
                const int gm = m0 + wm * (16 * MI) + half * 32 + row, gn = n0 + wn * 64 + q8 * 8;
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
        for (int i = 0; i < MI; ++i)
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
    if (d.dtype_b == FOCUS_FP8_E4M3) {
        // e4m3 weights: only this kernel consumes them (no env switch: there is no other path to fall back to)
        if (d.dtype_ab != FOCUS_BF16 || !d.b_scale || d.csA != 1 || d.rsB != 1 || d.K <= 0 || (d.K % BK) != 0) return false;
        if ((d.rsA & 7) || (d.bsA0 & 7) || (d.bsA1 & 7) || (d.csB & 15) || (d.bsB0 & 15) || (d.bsB1 & 15)) return false;
        if (!focus_aligned(d.A, 16) || !focus_aligned(d.B, 16)) return false;
    } else if (!enabled || !focus_gemm_mfma_nt_ok(d)) {
        return false;
    }
    if (d.dtype_c != FOCUS_BF16 || d.accumulate || d.csC != 1 || (d.rsC & 7) || (d.N & 7)) return false;
    if ((d.bsC0 & 7) || (d.bsC1 & 7)) return false;
    if (!focus_aligned(d.C, 16) || (d.residual && !focus_aligned(d.residual, 16)) || (d.aux && !focus_aligned(d.aux, 16)))
        return false;
    return true;
}

template <int WM, int WN, int MI, int EPI, bool BFP8>
static int launch_ws(const focus_gemm_desc& d, hipStream_t s) {
    // 4 loader waves: the LDS-DMA issue rate of the loaders, not the MFMA rate, bounds a K-step (~2.2k clocks per K-step
    // with 4 loaders x 12 pieces against 1k clocks of MFMA)
    constexpr int NLOAD = 4, BM = WM * MI * 16, BN = WN * 64;
    constexpr int STAGE = BM * 128 + BN * (BFP8 ? 64 : 128);
    constexpr int FIT = 160 * 1024 / STAGE;
    constexpr int NSTAGE = FIT >= 4 ? 4 : FIT;                    // ring depth: what fits the 160 KiB of LDS, at most 4
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    const int nbatch = d.batch0 * d.batch1;
    const size_t lds = (size_t)NSTAGE * STAGE;
    auto k = gemm_nt_ws_kernel<WM, WN, MI, EPI, NLOAD, NSTAGE, BFP8>;
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
    hipLaunchKernelGGL(k, grid, dim3(64 * (WM * WN + NLOAD)), lds, s, d, tiles_m, tiles_n, gm);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <int WM, int WN, int MI, bool BFP8>
static int launch_ws_epi(const focus_gemm_desc& d, hipStream_t s) {
    switch (d.epilogue) {
        case FOCUS_EPI_NONE: return launch_ws<WM, WN, MI, FOCUS_EPI_NONE, BFP8>(d, s);
        case FOCUS_EPI_GELU: return launch_ws<WM, WN, MI, FOCUS_EPI_GELU, BFP8>(d, s);
        case FOCUS_EPI_RELU: return launch_ws<WM, WN, MI, FOCUS_EPI_RELU, BFP8>(d, s);
        case FOCUS_EPI_TANH: return launch_ws<WM, WN, MI, FOCUS_EPI_TANH, BFP8>(d, s);
        case FOCUS_EPI_DGELU: return launch_ws<WM, WN, MI, FOCUS_EPI_DGELU, BFP8>(d, s);
        case FOCUS_EPI_DRELU: return launch_ws<WM, WN, MI, FOCUS_EPI_DRELU, BFP8>(d, s);
        case FOCUS_EPI_DTANH: return launch_ws<WM, WN, MI, FOCUS_EPI_DTANH, BFP8>(d, s);
        default: return FOCUS_ERR_SHAPE;
    }
}

static int g_tile_override = 0;
extern "C" int focus_gemm_tile_override(int bm) {
    if (bm != 0 && bm != 128 && bm != 160 && bm != 192 && bm != 193) return FOCUS_ERR_SHAPE;
    g_tile_override = bm;
    return FOCUS_OK;
}

template <bool BFP8>
static int dispatch_ws(const focus_gemm_desc& d, hipStream_t s) {
    const int nbatch = d.batch0 * d.batch1;
    // FOCUS_GEMM_TILE=128|160|192 or focus_gemm_tile_override() force a tile height (tuning); 193 = the round-2 192x256
    // form with 12 consumer waves
    static const int env_force = getenv("FOCUS_GEMM_TILE") ? atoi(getenv("FOCUS_GEMM_TILE")) : 0;
    const int force = g_tile_override ? g_tile_override : env_force;
    const int64_t tn256 = (d.N + 255) / 256;
    if (d.N >= 256 && nbatch == 1) {
        // tile height by modelled time = rounds of the 256 persistent workgroups x (rows per tile + a fixed cost per tile:
        // prologue, epilogue, ring refill ~ 40 rows' worth): 12552 x 768 -> 160 rows (237 tiles, one round),
        // 12552 x 2304 -> 160 (711 tiles, three rounds; 128: four), 12552 x 3072 -> 160 (four rounds; 128: five)
        int best = 0;
        double best_cost = 0;
        for (int bm = 128; bm <= 192; bm += 32) {
            const int64_t tiles = (int64_t)((d.M + bm - 1) / bm) * tn256;
            if (tiles < 128) continue;
            const double cost = (double)((tiles + 255) / 256) * (bm + 40);
            if (force ? bm == force : (!best || cost < best_cost - 1e-9)) { best = bm; best_cost = cost; }
        }
        if (force == 193 && !BFP8 && (int64_t)((d.M + 191) / 192) * tn256 >= 128) return launch_ws_epi<3, 4, 4, false>(d, s);
        if (best == 192) return launch_ws_epi<2, 4, 6, BFP8>(d, s);
        if (best == 160) return launch_ws_epi<2, 4, 5, BFP8>(d, s);
        if (best == 128 && (int64_t)((d.M + 127) / 128) * tn256 >= 192) return launch_ws_epi<2, 4, 4, BFP8>(d, s);
    }
    const int64_t t256 = (int64_t)((d.M + 255) / 256) * ((d.N + 127) / 128);
    const int64_t tw = (int64_t)((d.M + 127) / 128) * tn256;
    // narrow outputs (per-head products: N = head dim, batched over the heads): 4 consumers on a 256 x 64 tile, no MFMA
    // work on padding columns, and the 3-stage ring instead of the uniform kernel's one K-step in flight
    if (d.N <= 64 && d.M >= 2048 && d.epilogue == FOCUS_EPI_NONE && (int64_t)((d.M + 255) / 256) * nbatch >= 192)
        return launch_ws<4, 1, 4, FOCUS_EPI_NONE, BFP8>(d, s);
    if (d.N >= 256 && tw >= 192) return launch_ws_epi<2, 4, 4, BFP8>(d, s);   // 512-byte row segments of C
    if (d.M >= 256 && t256 >= 192) return launch_ws_epi<4, 2, 4, BFP8>(d, s);
    return FOCUS_ERR_SHAPE;   // too few tiles for one 8-consumer workgroup per CU: the caller uses the uniform kernel
}

int focus_gemm_mfma_ws(const focus_gemm_desc& d, hipStream_t s) {
    if (d.batch0 * d.batch1 > 65535) return FOCUS_ERR_SHAPE;
    if (d.dtype_b == FOCUS_FP8_E4M3) {
        // few tiles: the 128x256 instance still runs (under-filled); there is no uniform fp8 kernel
        const int rc = dispatch_ws<true>(d, s);
        if (rc != FOCUS_ERR_SHAPE) return rc;
        return d.N >= 256 ? launch_ws_epi<2, 4, 4, true>(d, s) : launch_ws_epi<4, 2, 4, true>(d, s);
    }
    return dispatch_ws<false>(d, s);
}
