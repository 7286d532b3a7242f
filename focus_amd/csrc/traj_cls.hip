// traj_cls.hip -- the cls query row of trajectory attention (attention.py:514-519): one query per (batch, head)
// attending over all N keys (cls + every patch token).  One 256-thread workgroup per (b, h); head dim 64.
// Replaces ~10 latency-bound M=1 GEMM/softmax launches per call by one kernel forward and one backward.
#include "focus_common.h"
#include "traj_internal.h"

namespace {

constexpr int HD = 64;

template <typename T>
__device__ __forceinline__ float dot64(const T* row, const float* q) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
        const f4 v = ld4<T>(row + c);
        s += v.x * q[c] + v.y * q[c + 1] + v.z * q[c + 2] + v.w * q[c + 3];
    }
    return s;
}

// dynamic LDS: prob[N] floats
template <typename T>
__global__ __launch_bounds__(1024) void cls_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ cls_out,
                                                      float* __restrict__ cls_lse, int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) float prob[];
    __shared__ float sq[HD], red[16], part[16][HD];
    const int bh = blockIdx.x, b = bh / heads, hh = bh % heads, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD);
    if (threadIdx.x < HD) sq[threadIdx.x] = ld<T>(base + threadIdx.x) * scale;
    __syncthreads();
    float m = -INFINITY;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const float l = dot64<T>(base + (int64_t)n * tok + C, sq);
        prob[n] = l;
        m = fmaxf(m, l);
    }
    m = block_max(m, red);
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 1024) { const float e = __expf(prob[n] - m); prob[n] = e; s += e; }
    s = block_sum(s, red);
    __syncthreads();
    const float inv = 1.f / s;
    const int dch = threadIdx.x & 63, pr = threadIdx.x >> 6;
    float acc = 0.f;
    for (int n = pr; n < N; n += 16) acc += prob[n] * ld<T>(base + (int64_t)n * tok + 2 * C + dch);
    part[pr][dch] = acc;
    __syncthreads();
    if (threadIdx.x < HD) {
        float t = 0.f;
        for (int y = 0; y < 16; ++y) t += part[y][threadIdx.x];
        st<T>(cls_out + (int64_t)b * C + hh * HD + threadIdx.x, t * inv);
        if (threadIdx.x == 0) cls_lse[bh] = m + __logf(s);
    }
}

// Backward, phase A (one workgroup per (b,h)): recompute the probabilities from the saved lse, form
// d logits (scale included) and dq0; prob and dlog go to scratch for phase B.
template <typename T>
__global__ __launch_bounds__(1024) void cls_bwd_a_kernel(const T* __restrict__ qkv, const float* __restrict__ cls_lse,
                                                         const T* __restrict__ dcls, T* __restrict__ dqkv,
                                                         float* __restrict__ prob_g, float* __restrict__ dlog_g, int N,
                                                         int heads) {
    extern __shared__ __attribute__((aligned(16))) float buf[];   // prob[N] | dlog[N]
    float* prob = buf;
    float* dlog = buf + N;
    __shared__ float sq[HD], sd[HD], red[16], part[16][HD];
    const int bh = blockIdx.x, b = bh / heads, hh = bh % heads, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD), lse = cls_lse[bh];
    if (threadIdx.x < HD) {
        sq[threadIdx.x] = ld<T>(base + threadIdx.x) * scale;
        sd[threadIdx.x] = ld<T>(dcls + (int64_t)b * C + hh * HD + threadIdx.x);
    }
    __syncthreads();
    float dot = 0.f;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const T* row = base + (int64_t)n * tok;
        const float a = __expf(dot64<T>(row + C, sq) - lse);
        const float da = dot64<T>(row + 2 * C, sd);
        prob[n] = a;
        dlog[n] = da;
        dot += a * da;
    }
    dot = block_sum(dot, red);
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 1024) {
        const float dl = scale * prob[n] * (dlog[n] - dot);
        dlog[n] = dl;
        prob_g[(int64_t)bh * N + n] = prob[n];
        dlog_g[(int64_t)bh * N + n] = dl;
    }
    __syncthreads();
    const int dch = threadIdx.x & 63, pr = threadIdx.x >> 6;
    float acc = 0.f;
    for (int n = pr; n < N; n += 16) acc += dlog[n] * ld<T>(base + (int64_t)n * tok + C + dch);
    part[pr][dch] = acc;
    __syncthreads();
    if (threadIdx.x < HD) {
        float t = 0.f;
        for (int y = 0; y < 16; ++y) t += part[y][threadIdx.x];
        st<T>(dqkv + (int64_t)b * N * tok + hh * HD + threadIdx.x, t);
    }
}

// Phase B (grid: 64-row strips x (b,h)): dK[n,:] (+)= dlog[n] * q0,  dV[n,:] (+)= prob[n] * dcls; token 0 is written
// plainly, tokens 1.. are read-modify-written (the patch kernels wrote them first).  16-byte accesses.
template <typename T>
__global__ __launch_bounds__(256) void cls_bwd_b_kernel(const T* __restrict__ qkv, const T* __restrict__ dcls,
                                                        const float* __restrict__ prob_g,
                                                        const float* __restrict__ dlog_g, T* __restrict__ dqkv, int N,
                                                        int heads) {
    const int bh = blockIdx.y, b = bh / heads, hh = bh % heads, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    for (int it = threadIdx.x; it < 64 * 16 * 2; it += 256) {
        const int which = it / (64 * 16), rem = it % (64 * 16), n = blockIdx.x * 64 + rem / 16, c = (rem % 16) * 4;
        if (n >= N) continue;
        const float w = which == 0 ? dlog_g[(int64_t)bh * N + n] : prob_g[(int64_t)bh * N + n];
        const T* src = which == 0 ? qkv + (int64_t)b * N * tok + hh * HD + c          // q0 (token 0, q part)
                                  : dcls + (int64_t)b * C + hh * HD + c;
        const f4 v = ld4<T>(src);
        T* dst = dqkv + ((int64_t)b * N + n) * tok + (which == 0 ? C : 2 * C) + hh * HD + c;
        f4 o = {w * v.x, w * v.y, w * v.z, w * v.w};
        if (n > 0) { const f4 old = ld4<T>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        st4<T>(dst, o);
    }
}

}  // namespace

// 2*N floats of dynamic LDS in the backward must stay under the 64 KiB default limit
bool focus_traj_cls_ok(int N, int d) { return d == HD && N <= 8000; }

int focus_traj_cls_fwd(const void* qkv, void* cls_out, float* cls_lse, int B, int N, int heads, int dtype, hipStream_t s) {
    const size_t lds = (size_t)N * sizeof(float);
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((cls_fwd_kernel<bf16_t>), dim3(B * heads), dim3(1024), lds, s, (const bf16_t*)qkv,
                           (bf16_t*)cls_out, cls_lse, N, heads);
    else
        hipLaunchKernelGGL((cls_fwd_kernel<float>), dim3(B * heads), dim3(1024), lds, s, (const float*)qkv,
                           (float*)cls_out, cls_lse, N, heads);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

int focus_traj_cls_bwd(const void* qkv, const float* cls_lse, const void* dcls, void* dqkv, float* scratch, int B,
                       int N, int heads, int dtype, hipStream_t s) {
    // scratch: 2 * B * heads * N floats (prob | dlog)
    const size_t lds = (size_t)2 * N * sizeof(float);
    float* prob_g = scratch;
    float* dlog_g = scratch + (size_t)B * heads * N;
    dim3 gb((N + 63) / 64, B * heads);
    if (dtype == FOCUS_BF16) {
        hipLaunchKernelGGL((cls_bwd_a_kernel<bf16_t>), dim3(B * heads), dim3(1024), lds, s, (const bf16_t*)qkv, cls_lse,
                           (const bf16_t*)dcls, (bf16_t*)dqkv, prob_g, dlog_g, N, heads);
        hipLaunchKernelGGL((cls_bwd_b_kernel<bf16_t>), gb, dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)dcls,
                           prob_g, dlog_g, (bf16_t*)dqkv, N, heads);
    } else {
        hipLaunchKernelGGL((cls_bwd_a_kernel<float>), dim3(B * heads), dim3(1024), lds, s, (const float*)qkv, cls_lse,
                           (const float*)dcls, (float*)dqkv, prob_g, dlog_g, N, heads);
        hipLaunchKernelGGL((cls_bwd_b_kernel<float>), gb, dim3(256), 0, s, (const float*)qkv, (const float*)dcls, prob_g,
                           dlog_g, (float*)dqkv, N, heads);
    }
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
