// gemm_generic.hip -- strided, batched GEMM with fused epilogue, any layout, fp32 math on the VALU.
// This is the fp32 ("TRAIN.MIXED_PRECISION False") path and the fallback for operand layouts the
// MFMA kernels (gemm_mfma.hip) do not take.  64x64 output tile, BK=16, 256 threads, 4x4 per thread.
#include "focus_common.h"
#include "gemm_internal.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16;

template <typename TC>
__device__ __forceinline__ void epilogue_store(const focus_gemm_desc& d, TC* C, const TC* R, TC* X, int64_t off,
                                               int n, float acc) {
    float v = d.alpha * acc;
    if (d.bias) v += d.bias[n];
    switch (d.epilogue) {
        case FOCUS_EPI_GELU:
            if (X) st<TC>(X + off, v);
            v = gelu_erf(v);
            break;
        case FOCUS_EPI_RELU: v = fmaxf(v, 0.f); break;
        case FOCUS_EPI_TANH: v = tanhf(v); break;
        case FOCUS_EPI_DGELU: v *= dgelu_erf(ld<TC>(X + off)); break;
        case FOCUS_EPI_DRELU: v = ld<TC>(X + off) > 0.f ? v : 0.f; break;
        case FOCUS_EPI_DTANH: { float y = ld<TC>(X + off); v *= (1.f - y * y); } break;
        default: break;
    }
    if (R) v += ld<TC>(R + off);
    if (d.accumulate) v += ld<TC>(C + off);
    st<TC>(C + off, v);
}

template <typename TA, typename TC>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const focus_gemm_desc d) {
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int z = blockIdx.z;
    const int b0 = z / d.batch1, b1 = z % d.batch1;
    const TA* A = static_cast<const TA*>(d.A) + b0 * d.bsA0 + b1 * d.bsA1;
    const TA* B = static_cast<const TA*>(d.B) + b0 * d.bsB0 + b1 * d.bsB1;
    const int64_t coff = b0 * d.bsC0 + b1 * d.bsC1;
    TC* C = static_cast<TC*>(d.C) + coff;
    const TC* R = d.residual ? static_cast<const TC*>(d.residual) + coff : nullptr;
    TC* X = d.aux ? static_cast<TC*>(d.aux) + coff : nullptr;

    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;  // 16 x 16 threads, each 4 rows x 4 cols
    const bool a_kfast = (d.csA == 1), b_nfast = (d.csB == 1);

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < d.K; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            int m, k;
            if (a_kfast) { m = e >> 4; k = e & 15; } else { m = e & 63; k = e >> 6; }
            const int gm = m0 + m, gk = k0 + k;
            As[k][m] = (gm < d.M && gk < d.K) ? ld<TA>(A + gm * d.rsA + gk * d.csA) : 0.f;
            int n, kb;
            if (b_nfast) { n = e & 63; kb = e >> 6; } else { n = e >> 4; kb = e & 15; }
            const int gn = n0 + n, gkb = k0 + kb;
            Bs[kb][n] = (gn < d.N && gkb < d.K) ? ld<TA>(B + gkb * d.rsB + gn * d.csB) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= d.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= d.N) continue;
            epilogue_store<TC>(d, C, R, X, gm * d.rsC + gn * d.csC, gn, acc[i][j]);
        }
    }
}

}  // namespace

int focus_gemm_generic(const focus_gemm_desc& d, hipStream_t s) {
    dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.batch0 * d.batch1);
    if (grid.y > 65535 || grid.z > 65535) return FOCUS_ERR_SHAPE;
    if (d.dtype_ab == FOCUS_F32 && d.dtype_c == FOCUS_F32)
        hipLaunchKernelGGL((gemm_generic_kernel<float, float>), grid, dim3(256), 0, s, d);
    else if (d.dtype_ab == FOCUS_BF16 && d.dtype_c == FOCUS_BF16)
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, d);
    else if (d.dtype_ab == FOCUS_BF16 && d.dtype_c == FOCUS_F32)
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, float>), grid, dim3(256), 0, s, d);
    else if (d.dtype_ab == FOCUS_F32 && d.dtype_c == FOCUS_BF16)
        hipLaunchKernelGGL((gemm_generic_kernel<float, bf16_t>), grid, dim3(256), 0, s, d);
    else
        return FOCUS_ERR_DTYPE;
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
