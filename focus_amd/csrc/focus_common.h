// focus_common.h -- shared device/host helpers for libfocus_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/focus_amd.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

// ---- storage <-> fp32 ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// Plain cast keeps NaNs (MI355X_MICROARCH.md "Correctness boundaries"); hipcc emits v_cvt_pk_bf16_f32.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// 4-wide vector access (16 B for fp32, 8 B for bf16); pointers must be aligned accordingly.
struct f4 { float x, y, z, w; };
template <typename T> __device__ __forceinline__ f4 ld4(const T* p);
template <> __device__ __forceinline__ f4 ld4<float>(const float* p) {
    float4 v = *reinterpret_cast<const float4*>(p);
    return {v.x, v.y, v.z, v.w};
}
template <> __device__ __forceinline__ f4 ld4<bf16_t>(const bf16_t* p) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    return {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
            __uint_as_float(v.y & 0xffff0000u)};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f4 v) {
    *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f4 v) {
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    o.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = o;
}

// ---- LDS-DMA issued from inline asm ----------------------------------------------------------------
// global_load_lds_dwordx4 hidden from hipcc: with the builtin the compiler cannot tell which later LDS accesses alias
// the DMA in flight and drains s_waitcnt vmcnt(0) before them (measured in traj_space_fwd: every frame's prefetch was
// waited for right after it was issued).  The statement is absent from the compiler's vmcnt bookkeeping: the caller
// counts it (s_waitcnt vmcnt(N)), then a barrier, then the ds_read (cdna_hip_programming.md, inline-asm rules).
// lds_dst: wave-uniform LDS byte address of the 1 KiB piece; lane l writes bytes [16 l, 16 l + 16) of it.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// 4 bytes per lane: lane l writes bytes [4 l, 4 l + 4) of the 256-B piece at lds_dst
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ---- 8 consecutive elements as floats: one 16-byte access for bf16, two for fp32 (16-byte aligned address) ----
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&f)[8]);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&f)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&f)[8]);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&f)[8]) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(f[0]) | ((uint32_t)f32_to_bf16(f[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(f[2]) | ((uint32_t)f32_to_bf16(f[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(f[4]) | ((uint32_t)f32_to_bf16(f[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(f[6]) | ((uint32_t)f32_to_bf16(f[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&f)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
// ---- wave (64 lanes) and block reductions --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- activations -------------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, branch-free: one rcp, one exp2, five fma) instead of ocml's
// erff (two polynomial ranges, both executed under divergence): the GELU epilogues of the fc1 / d(fc1) GEMMs spend
// their time here.  *e_out = exp(-z*z), which GELU' needs anyway.
__device__ __forceinline__ float erf_as(float z, float* e_out) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    const float e = __builtin_amdgcn_exp2f(az * az * -1.4426950408889634f);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    *e_out = e;
    return copysignf(fmaf(-p * t, e, 1.0f), z);
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f, &e));
}
__device__ __forceinline__ float dgelu_erf(float x) {
    float e;                                                   // exp(-x*x/2)
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f, &e));
    return fmaf(x * 0.3989422804014327f, e, cdf);
}

// ---- host side ----------------------------------------------------------------------------------
#define FOCUS_CHECK_LAUNCH()                                  \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return FOCUS_ERR_LAUNCH; \
    } while (0)

static inline bool focus_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline size_t focus_esize(int dtype) { return dtype == FOCUS_BF16 ? 2 : 4; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
