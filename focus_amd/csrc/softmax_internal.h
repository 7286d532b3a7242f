#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
int focus_softmax_fwd_lse(const void* x, void* y, float* lse, int64_t rows, int L, int64_t stride, float scale,
                          int dtype, hipStream_t s);
int focus_softmax_from_lse(const void* x, void* y, const float* lse, int64_t rows, int L, int64_t stride,
                           float scale, int dtype, hipStream_t s);
