// gemm_mfma_small.hip -- bf16 MFMA GEMM for SMALL row counts:  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias) + residual
//
// The recurrent part of STEVE (steve.py:68-93: q projection, GRU gates, slot MLP, predictor) and ORViT's motion stream
// apply Linear layers to B*K = 352 (resp. 256) rows: 50-300 MFLOP per product.  The tiled kernels spend 12-57 us on
// them -- 12..15 workgroups walking a serial K loop with one barrier and one HBM/L2 round trip per 64-deep step -- and
// there are ~760 such launches per STEVE step.  Here the product is latency-shaped instead:
//   * workgroup = one 32 x 32 output tile, 4 waves; wave w takes every 4th 32-deep K step (in-workgroup split-K), so a
//     wave has at most 6 steps for K <= 768 and issues ALL of its operand loads at once: one memory round trip;
//   * operands go global -> registers directly in MFMA layout (16 B per lane), no LDS staging, no K loop barriers;
//   * the four partial tiles meet in LDS (16 KB), wave w finishes sub-tile (w >> 1, w & 1) with the same epilogue
//     semantics as the tiled kernels (alpha, bias, GELU/ReLU/tanh and their derivative forms, aux, residual).
// M x N / 1024 workgroups (66 .. 264 for the STEVE shapes): every CU gets work, nothing is serial.
#include "focus_common.h"
#include "gemm_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TS = 32;          // output tile (rows and columns)
constexpr int MAXSTEPS = 12;    // K steps of 32 per wave held in registers per pass: 4 * 12 * 32 = 1536 of K per pass

template <int EPI, typename TC, int NS>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(const focus_gemm_desc d, int tiles_n, int npass) {
    __shared__ __attribute__((aligned(16))) float part[4][4][64][4];       // [wave][sub-tile][lane][4]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * TS, n0 = tn * TS;
    const int z = blockIdx.y;
    const int b0 = z / d.batch1, b1 = z % d.batch1;
    const bf16_t* A = static_cast<const bf16_t*>(d.A) + b0 * d.bsA0 + b1 * d.bsA1;
    const bf16_t* B = static_cast<const bf16_t*>(d.B) + b0 * d.bsB0 + b1 * d.bsB1;
    const int64_t coff = b0 * d.bsC0 + b1 * d.bsC1;
    TC* C = static_cast<TC*>(d.C) + coff;
    const TC* R = d.residual ? static_cast<const TC*>(d.residual) + coff : nullptr;
    TC* X = d.aux ? static_cast<TC*>(d.aux) + coff : nullptr;
    const int frow = lane & 15, fq = lane >> 4;
    const int nsteps = d.K / 32;                                  // steps w, w + 4, ... belong to wave w
    // all operand fragments of this wave: one round trip per pass of NS steps (one pass for K <= 128 NS; the motion stream's
    // K = 3072 products take two: 192 workgroups x two round trips instead of 12 workgroups walking 48 barrier-locked steps)
    const bf16_t* arow[2];
    const bf16_t* brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = A + (int64_t)min(m0 + 16 * i + frow, d.M - 1) * d.rsA + 8 * fq;
        brow[i] = B + (int64_t)min(n0 + 16 * i + frow, d.N - 1) * d.csB + 8 * fq;
    }
    // the epilogue's own operands (bias, residual, saved activation of the sub-tile this wave finishes) are requested now,
    // together with the fragments: after the LDS combine they would be a second, fully exposed round trip
    const int gm = m0 + 16 * (w >> 1) + frow, gn = n0 + 16 * (w & 1) + 4 * fq;
    const bool live = gm < d.M && gn < d.N;
    const int64_t off = (int64_t)gm * d.rsC + gn;                 // csC == 1 (checked by the dispatcher)
    const bool full = gn + 3 < d.N && ((d.rsC & 3) == 0);
    float xs[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (int r = 0; r < 4; ++r) {
            if (gn + r >= d.N) break;
            if (d.bias) bs[r] = d.bias[gn + r];
        }
        if constexpr (EPI >= FOCUS_EPI_DGELU) {
            if (full) { const f4 xa = ld4<TC>(X + off); xs[0] = xa.x; xs[1] = xa.y; xs[2] = xa.z; xs[3] = xa.w; }
            else for (int r = 0; r < 4 && gn + r < d.N; ++r) xs[r] = ld<TC>(X + off + r);
        }
        if (R) {
            if (full) { const f4 rr = ld4<TC>(R + off); rs[0] = rr.x; rs[1] = rr.y; rs[2] = rr.z; rs[3] = rr.w; }
            else for (int r = 0; r < 4 && gn + r < d.N; ++r) rs[r] = ld<TC>(R + off + r);
        }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < npass; ++pass) {
        bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ks = min(w + 4 * (pass * NS + s), nsteps - 1) * 32;   // steps past the end re-read the last one (not used)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[s][i] = *reinterpret_cast<const bf16x8*>(arow[i] + ks);
                fb[s][i] = *reinterpret_cast<const bf16x8*>(brow[i] + ks);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (w + 4 * (pass * NS + s) < nsteps) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)                   // "swapped": the lane ends with 4 consecutive columns of one row
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float4*>(&part[w][2 * i + j][lane][0]) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();
    // wave w finishes sub-tile (i, j) = (w >> 1, w & 1)
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
        const float4 p = *reinterpret_cast<const float4*>(&part[ww][w][lane][0]);
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    }
    if (!live) return;
    float pre[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = d.alpha * v[r] + bs[r];
        pre[r] = t;
        if constexpr (EPI == FOCUS_EPI_GELU) t = gelu_erf(t);
        else if constexpr (EPI == FOCUS_EPI_RELU) t = fmaxf(t, 0.f);
        else if constexpr (EPI == FOCUS_EPI_TANH) t = tanhf(t);
        else if constexpr (EPI == FOCUS_EPI_DGELU) t *= dgelu_erf(xs[r]);
        else if constexpr (EPI == FOCUS_EPI_DRELU) t = xs[r] > 0.f ? t : 0.f;
        else if constexpr (EPI == FOCUS_EPI_DTANH) t *= (1.f - xs[r] * xs[r]);
        v[r] = t + rs[r];
    }
    if (full) {
        if constexpr (EPI == FOCUS_EPI_GELU) { if (X) st4<TC>(X + off, (f4){pre[0], pre[1], pre[2], pre[3]}); }
        st4<TC>(C + off, (f4){v[0], v[1], v[2], v[3]});
    } else {
        for (int r = 0; r < 4; ++r) {
            if (gn + r >= d.N) break;
            if constexpr (EPI == FOCUS_EPI_GELU) { if (X) st<TC>(X + off + r, pre[r]); }
            st<TC>(C + off + r, v[r]);
        }
    }
}

template <int EPI, typename TC>
void launch_small(const focus_gemm_desc& d, hipStream_t s) {
    const int tiles_m = (d.M + TS - 1) / TS, tiles_n = (d.N + TS - 1) / TS;
    const dim3 grid(tiles_m * tiles_n, d.batch0 * d.batch1);
    const int per_wave = (d.K / 32 + 3) / 4;
    if (per_wave <= 2) hipLaunchKernelGGL((gemm_nt_small_kernel<EPI, TC, 2>), grid, dim3(256), 0, s, d, tiles_n, 1);
    else if (per_wave <= 6) hipLaunchKernelGGL((gemm_nt_small_kernel<EPI, TC, 6>), grid, dim3(256), 0, s, d, tiles_n, 1);
    else hipLaunchKernelGGL((gemm_nt_small_kernel<EPI, TC, MAXSTEPS>), grid, dim3(256), 0, s, d, tiles_n,
                            (per_wave + MAXSTEPS - 1) / MAXSTEPS);
}

template <typename TC>
int launch_small_epi(const focus_gemm_desc& d, hipStream_t s) {
    switch (d.epilogue) {
        case FOCUS_EPI_NONE: launch_small<FOCUS_EPI_NONE, TC>(d, s); break;
        case FOCUS_EPI_GELU: launch_small<FOCUS_EPI_GELU, TC>(d, s); break;
        case FOCUS_EPI_RELU: launch_small<FOCUS_EPI_RELU, TC>(d, s); break;
        case FOCUS_EPI_TANH: launch_small<FOCUS_EPI_TANH, TC>(d, s); break;
        case FOCUS_EPI_DGELU: launch_small<FOCUS_EPI_DGELU, TC>(d, s); break;
        case FOCUS_EPI_DRELU: launch_small<FOCUS_EPI_DRELU, TC>(d, s); break;
        case FOCUS_EPI_DTANH: launch_small<FOCUS_EPI_DTANH, TC>(d, s); break;
        default: return FOCUS_ERR_SHAPE;
    }
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

// Shapes this kernel takes: few rows, a reduction short enough for the register-resident operands, dense C rows.
bool focus_gemm_mfma_small_ok(const focus_gemm_desc& d) {
    static const bool enabled = !(getenv("FOCUS_GEMM_SMALL") && atoi(getenv("FOCUS_GEMM_SMALL")) == 0);
    if (!enabled || !focus_gemm_mfma_nt_ok(d)) return false;      // bf16, both operands K-contiguous, aligned
    static const int passes = getenv("FOCUS_GEMM_SMALL_PASSES") ? std::max(1, atoi(getenv("FOCUS_GEMM_SMALL_PASSES"))) : 3;
    if (d.M > 1024 || d.K > passes * 4 * MAXSTEPS * 32 || (d.K & 31) || d.accumulate || d.csC != 1) return false;
    if (d.dtype_c != FOCUS_BF16 && d.dtype_c != FOCUS_F32) return false;
    const int64_t tiles = (int64_t)((d.M + TS - 1) / TS) * ((d.N + TS - 1) / TS);
    if (tiles > 4096 || d.batch0 * d.batch1 > 65535) return false;
    if ((d.rsC & 3) || (d.bsC0 & 3) || (d.bsC1 & 3)) return false;
    if (!focus_aligned(d.C, 16) || (d.residual && !focus_aligned(d.residual, 16)) || (d.aux && !focus_aligned(d.aux, 16)))
        return false;
    return true;
}

int focus_gemm_mfma_small(const focus_gemm_desc& d, hipStream_t s) {
    if (d.dtype_c == FOCUS_BF16) return launch_small_epi<bf16_t>(d, s);
    return launch_small_epi<float>(d, s);
}
