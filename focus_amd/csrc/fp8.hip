// fp8.hip -- OCP FP8 E4M3 (e4m3fn) working copies of the fp32 master weights with ONE scale per tensor
// (BASELINE configs[4]: "fp8 MFMA weights", SURVEY 8(d) config 5: fp8 weights for the Linear GEMMs, bf16 activations;
// reference call sites of the weights: attention.py:506,536-537,555, common.py:26-34, orvit.py:59-64,98-100).
//   scale = amax / 448,   code = RNE_e4m3(w * (448 / amax)),   w ~= decode(code) * scale
// Two launches over all tensors of a model (after a memset of the amax words):
//   fp8_amax_kernel   : amax[item] = max |w|  (atomicMax on the fp32 bit pattern: non-negative floats order like uints)
//   fp8_quant_kernel  : row-major codes [rows, cols] (B operand of the forward NT GEMM) and transposed codes [cols, rows]
//                       (B operand of the dX GEMM) through a 64 x 64 LDS tile; scale[item] for the GEMM epilogue.
// The conversion is gfx950's v_cvt_pk_fp8_f32 (OCP encoding on this chip, MI355X_MICROARCH.md "FP8 (e4m3/e5m2)": OCP
// e4m3fn, not MI300X fnuz); tests/test_gpu_fp8.py checks every code against oracle/fp8.py.
#include "focus_common.h"

namespace {

__global__ __launch_bounds__(256) void fp8_amax_kernel(const focus_fp8_item* __restrict__ items, uint32_t* __restrict__ amax) {
    __shared__ float red[16];
    const focus_fp8_item it = items[blockIdx.y];
    const int tiles_c = (it.cols + 63) >> 6, tiles_r = (it.rows + 63) >> 6;
    if ((int)blockIdx.x >= tiles_r * tiles_c) return;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const int t = threadIdx.x, lr = t >> 4, lc = (t & 15) * 4;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = tr * 64 + lr + 16 * k, c = tc * 64 + lc;
        if (r < it.rows && c < it.cols) {
            const float4 v = *reinterpret_cast<const float4*>(it.src + (int64_t)r * it.cols + c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    m = block_max(m, red);
    if (t == 0) atomicMax(amax + blockIdx.y, __float_as_uint(m));
}

__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);      // bytes 0, 1
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);       // bytes 2, 3
    return (uint32_t)v;
}

__global__ __launch_bounds__(256) void fp8_quant_kernel(const focus_fp8_item* __restrict__ items, const uint32_t* __restrict__ amax) {
    __shared__ uint8_t tile[64][68];
    const focus_fp8_item it = items[blockIdx.y];
    const int tiles_c = (it.cols + 63) >> 6, tiles_r = (it.rows + 63) >> 6;
    if ((int)blockIdx.x >= tiles_r * tiles_c) return;
    const float am = __uint_as_float(amax[blockIdx.y]);
    const bool ok = am > 0.f && am < INFINITY;
    const float inv = ok ? 448.0f / am : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *it.scale = ok ? am / 448.0f : 1.0f;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const int t = threadIdx.x, lr = t >> 4, lc = (t & 15) * 4;
    uint8_t* dst = static_cast<uint8_t*>(it.dst);
    uint8_t* dstT = static_cast<uint8_t*>(it.dstT);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = tr * 64 + lr + 16 * k, c = tc * 64 + lc;
        uint32_t o = 0;
        if (r < it.rows && c < it.cols) {
            const float4 v = *reinterpret_cast<const float4*>(it.src + (int64_t)r * it.cols + c);
            o = pack4_e4m3(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
            if (dst) *reinterpret_cast<uint32_t*>(dst + (int64_t)r * it.cols + c) = o;
        }
        *reinterpret_cast<uint32_t*>(&tile[lr + 16 * k][lc]) = o;
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int oc = tc * 64 + lr + 16 * k, orr = tr * 64 + lc;      // output row = source column
        if (oc < it.cols && orr < it.rows) {
            const uint32_t o = (uint32_t)tile[lc + 0][lr + 16 * k] | ((uint32_t)tile[lc + 1][lr + 16 * k] << 8) |
                               ((uint32_t)tile[lc + 2][lr + 16 * k] << 16) | ((uint32_t)tile[lc + 3][lr + 16 * k] << 24);
            *reinterpret_cast<uint32_t*>(dstT + (int64_t)oc * it.rows + orr) = o;
        }
    }
}

}  // namespace

extern "C" int focus_fp8_refresh(const focus_fp8_item* items, int n_items, int max_rows, int max_cols, void* amax_scratch,
                                 void* stream) {
    if (!items || !amax_scratch) return FOCUS_ERR_NULL;
    if (n_items <= 0) return FOCUS_OK;
    if (n_items > 65535 || max_rows <= 0 || max_cols <= 0) return FOCUS_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(amax_scratch, 0, (size_t)n_items * 4, s) != hipSuccess) return FOCUS_ERR_LAUNCH;
    const int tiles = ((max_rows + 63) / 64) * ((max_cols + 63) / 64);
    hipLaunchKernelGGL(fp8_amax_kernel, dim3(tiles, n_items), dim3(256), 0, s, items, (uint32_t*)amax_scratch);
    FOCUS_CHECK_LAUNCH();
    hipLaunchKernelGGL(fp8_quant_kernel, dim3(tiles, n_items), dim3(256), 0, s, items, (const uint32_t*)amax_scratch);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
