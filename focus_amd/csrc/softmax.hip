// softmax.hip -- row softmax over segments (one wave per row), forward with optional log-sum-exp output
// and backward.  y = softmax(scale * x);  dx = scale * y * (dy - sum(dy * y)).
#include "focus_common.h"
#include "softmax_internal.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          float* __restrict__ lse, int64_t rows, int L,
                                                          int64_t stride, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * stride;
    T* yr = y + row * stride;
    float m = -INFINITY;
    for (int i = lane; i < L; i += 64) m = fmaxf(m, scale * ld<T>(xr + i));
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += __expf(scale * ld<T>(xr + i) - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int i = lane; i < L; i += 64) st<T>(yr + i, __expf(scale * ld<T>(xr + i) - m) * inv);
    if (lse && lane == 0) lse[row] = m + __logf(s);
}

// Causal rows: row r = (.., i) with i = r % period sees keys 0 .. i; the masked tail is written as exact zeros (the
// reference fills it with -inf before its softmax, transformer.py:41-44), so focus_softmax_bwd needs no mask.
template <typename T>
__global__ __launch_bounds__(256) void softmax_causal_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows,
                                                                 int L, int64_t stride, int period, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int valid = min(L, (int)(row % period) + 1);
    const T* xr = x + row * stride;
    T* yr = y + row * stride;
    float m = -INFINITY;
    for (int i = lane; i < valid; i += 64) m = fmaxf(m, scale * ld<T>(xr + i));
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < valid; i += 64) s += __expf(scale * ld<T>(xr + i) - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int i = lane; i < L; i += 64) st<T>(yr + i, i < valid ? __expf(scale * ld<T>(xr + i) - m) * inv : 0.f);
}

// Recompute probabilities from saved log-sum-exp: y = exp(scale*x - lse).
template <typename T>
__global__ __launch_bounds__(256) void softmax_from_lse_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                               const float* __restrict__ lse, int64_t rows, int L,
                                                               int64_t stride, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float l = lse[row];
    for (int i = lane; i < L; i += 64) st<T>(y + row * stride + i, __expf(scale * ld<T>(x + row * stride + i) - l));
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                          T* __restrict__ dx, int64_t rows, int L, int64_t stride,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* dr = dy + row * stride;
    const T* yr = y + row * stride;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += ld<T>(dr + i) * ld<T>(yr + i);
    s = wave_sum(s);
    for (int i = lane; i < L; i += 64) st<T>(dx + row * stride + i, scale * ld<T>(yr + i) * (ld<T>(dr + i) - s));
}

}  // namespace

int focus_softmax_fwd_lse(const void* x, void* y, float* lse, int64_t rows, int L, int64_t stride, float scale,
                          int dtype, hipStream_t s) {
    if (rows <= 0 || L <= 0) return FOCUS_OK;
    const int64_t nb = cdiv64(rows, 4);
    if (nb > 0x7fffffff) return FOCUS_ERR_SHAPE;
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)x,
                           (bf16_t*)y, lse, rows, L, stride, scale);
    else
        hipLaunchKernelGGL((softmax_fwd_kernel<float>), dim3((unsigned)nb), dim3(256), 0, s, (const float*)x,
                           (float*)y, lse, rows, L, stride, scale);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

int focus_softmax_from_lse(const void* x, void* y, const float* lse, int64_t rows, int L, int64_t stride,
                           float scale, int dtype, hipStream_t s) {
    if (rows <= 0 || L <= 0) return FOCUS_OK;
    const int64_t nb = cdiv64(rows, 4);
    if (nb > 0x7fffffff) return FOCUS_ERR_SHAPE;
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((softmax_from_lse_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, s,
                           (const bf16_t*)x, (bf16_t*)y, lse, rows, L, stride, scale);
    else
        hipLaunchKernelGGL((softmax_from_lse_kernel<float>), dim3((unsigned)nb), dim3(256), 0, s, (const float*)x,
                           (float*)y, lse, rows, L, stride, scale);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_softmax_fwd(const void* x, void* y, int64_t rows, int L, int64_t stride, float scale,
                                 int dtype, void* stream) {
    if (!x || !y) return FOCUS_ERR_NULL;
    return focus_softmax_fwd_lse(x, y, nullptr, rows, L, stride, scale, dtype, (hipStream_t)stream);
}

extern "C" int focus_softmax_causal_fwd(const void* x, void* y, int64_t rows, int L, int64_t stride, int period, float scale,
                                        int dtype, void* stream) {
    if (!x || !y) return FOCUS_ERR_NULL;
    if (rows <= 0 || L <= 0) return FOCUS_OK;
    if (period <= 0) return FOCUS_ERR_SHAPE;
    const int64_t nb = cdiv64(rows, 4);
    if (nb > 0x7fffffff) return FOCUS_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((softmax_causal_fwd_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)x,
                           (bf16_t*)y, rows, L, stride, period, scale);
    else
        hipLaunchKernelGGL((softmax_causal_fwd_kernel<float>), dim3((unsigned)nb), dim3(256), 0, s, (const float*)x,
                           (float*)y, rows, L, stride, period, scale);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_softmax_bwd(const void* dy, const void* y, void* dx, int64_t rows, int L, int64_t stride,
                                 float scale, int dtype, void* stream) {
    if (!dy || !y || !dx) return FOCUS_ERR_NULL;
    if (rows <= 0 || L <= 0) return FOCUS_OK;
    const int64_t nb = cdiv64(rows, 4);
    if (nb > 0x7fffffff) return FOCUS_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)dy,
                           (const bf16_t*)y, (bf16_t*)dx, rows, L, stride, scale);
    else
        hipLaunchKernelGGL((softmax_bwd_kernel<float>), dim3((unsigned)nb), dim3(256), 0, s, (const float*)dy,
                           (const float*)y, (float*)dx, rows, L, stride, scale);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
