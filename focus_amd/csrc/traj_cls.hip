// traj_cls.hip -- the cls query row of trajectory attention (attention.py:514-519): one query per (batch, head)
// attending over all N keys (cls + every patch token); head dim 64.
// Replaces ~10 latency-bound M=1 GEMM/softmax launches per call by two small kernels forward and two backward.
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

// The N keys of one (b,h) are split over NS workgroups (flash-decoding style): with one workgroup per (b,h) only 96
// of the 256 CUs had work and each streamed its 400 KB of K/V rows at latency (39 / 58 us per call).
constexpr int NS = 8;            // key splits per (b,h)
constexpr int FW = 2 + HD;       // forward partial: max, sum, out[64]
constexpr int BW = 1 + 2 * HD;   // backward partial: sum a*da, sum a*da*K[64], sum a*K[64]

// Forward, part: split sp of (b,h) -> scratch[bh][sp] = (m, sum exp(l - m), sum exp(l - m) * V[n,:])
template <typename T>
__global__ __launch_bounds__(256) void cls_fwd_part_kernel(const T* __restrict__ qkv, float* __restrict__ scratch, int N,
                                                          int heads) {
    __shared__ float sq[HD], red[4], prob[256], part[4][HD];
    const int sp = blockIdx.x, bh = blockIdx.y, b = bh / heads, hh = bh % heads, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD);
    const int chunk = (N + NS - 1) / NS, n0 = sp * chunk, n1 = min(N, n0 + chunk);
    if (threadIdx.x < HD) sq[threadIdx.x] = ld<T>(base + threadIdx.x) * scale;
    __syncthreads();
    const int dch = threadIdx.x & 63, pr = threadIdx.x >> 6;
    float m = -INFINITY, ssum = 0.f, acc = 0.f;
    for (int c0 = n0; c0 < n1; c0 += 256) {                   // (one pass for chunk <= 256 keys)
        const int n = c0 + threadIdx.x;
        const float l = n < n1 ? dot64<T>(base + (int64_t)n * tok + C, sq) : -INFINITY;
        const float mc = fmaxf(m, block_max(l, red));
        __syncthreads();
        const float e = n < n1 ? __expf(l - mc) : 0.f;
        prob[threadIdx.x] = e;
        const float resc = __expf(m - mc);                    // 0 on the first pass (m = -inf)
        ssum = ssum * resc + block_sum(e, red);
        __syncthreads();
        acc *= resc;
        const int cnt = min(256, n1 - c0);
        for (int j = pr; j < cnt; j += 4) acc += prob[j] * ld<T>(base + (int64_t)(c0 + j) * tok + 2 * C + dch);
        m = mc;
        __syncthreads();
    }
    part[pr][dch] = acc;
    __syncthreads();
    float* out = scratch + ((int64_t)bh * NS + sp) * FW;
    if (threadIdx.x < HD) out[2 + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (threadIdx.x == 0) { out[0] = m; out[1] = ssum; }
}

// Forward, combine: one 64-thread workgroup per (b,h)
template <typename T>
__global__ __launch_bounds__(64) void cls_fwd_comb_kernel(const float* __restrict__ scratch, T* __restrict__ cls_out,
                                                         float* __restrict__ cls_lse, int heads) {
    const int bh = blockIdx.x, b = bh / heads, hh = bh % heads, C = heads * HD;
    const float* p = scratch + (int64_t)bh * NS * FW;
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < NS; ++k) M = fmaxf(M, p[k * FW]);
    float S = 0.f, o = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float w = __expf(p[k * FW] - M);               // empty splits carry m = -inf -> weight 0
        S += p[k * FW + 1] * w;
        o += p[k * FW + 2 + threadIdx.x] * w;
    }
    st<T>(cls_out + (int64_t)b * C + hh * HD + threadIdx.x, o / S);
    if (threadIdx.x == 0) cls_lse[bh] = M + __logf(S);
}

// Backward, phase A (split sp of (b,h)): probabilities from the saved lse, da[n] = dcls . V[n]; both go to scratch
// for phase B together with this split's partial sums  sum a*da,  sum a*da*K[n,:],  sum a*K[n,:].
template <typename T>
__global__ __launch_bounds__(256) void cls_bwd_a_kernel(const T* __restrict__ qkv, const float* __restrict__ cls_lse,
                                                        const T* __restrict__ dcls, float* __restrict__ prob_g,
                                                        float* __restrict__ da_g, float* __restrict__ parts, int N,
                                                        int heads) {
    __shared__ float sq[HD], sd[HD], red[4], pa[256], pd[256], part[2][4][HD];
    const int sp = blockIdx.x, bh = blockIdx.y, b = bh / heads, hh = bh % heads, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD), lse = cls_lse[bh];
    const int chunk = (N + NS - 1) / NS, n0 = sp * chunk, n1 = min(N, n0 + chunk);
    if (threadIdx.x < HD) {
        sq[threadIdx.x] = ld<T>(base + threadIdx.x) * scale;
        sd[threadIdx.x] = ld<T>(dcls + (int64_t)b * C + hh * HD + threadIdx.x);
    }
    __syncthreads();
    const int dch = threadIdx.x & 63, pr = threadIdx.x >> 6;
    float dot = 0.f, u = 0.f, wv = 0.f;
    for (int c0 = n0; c0 < n1; c0 += 256) {
        const int n = c0 + threadIdx.x;
        float a = 0.f, da = 0.f;
        if (n < n1) {
            const T* row = base + (int64_t)n * tok;
            a = __expf(dot64<T>(row + C, sq) - lse);
            da = dot64<T>(row + 2 * C, sd);
            prob_g[(int64_t)bh * N + n] = a;
            da_g[(int64_t)bh * N + n] = da;
        }
        dot += a * da;
        pa[threadIdx.x] = a;
        pd[threadIdx.x] = a * da;
        __syncthreads();
        const int cnt = min(256, n1 - c0);
        for (int j = pr; j < cnt; j += 4) {
            const float k = ld<T>(base + (int64_t)(c0 + j) * tok + C + dch);
            u += pd[j] * k;
            wv += pa[j] * k;
        }
        __syncthreads();
    }
    dot = block_sum(dot, red);
    part[0][pr][dch] = u;
    part[1][pr][dch] = wv;
    __syncthreads();
    float* out = parts + ((int64_t)bh * NS + sp) * BW;
    if (threadIdx.x < 2 * HD) {
        const int which = threadIdx.x >> 6, d = threadIdx.x & 63;
        out[1 + which * HD + d] = part[which][0][d] + part[which][1][d] + part[which][2][d] + part[which][3][d];
    }
    if (threadIdx.x == 0) out[0] = dot;
}

// Phase B (grid: 64-row strips x (b,h)): dl[n] = scale * a[n] * (da[n] - dot);  dK[n,:] (+)= dl[n] * q0,
// dV[n,:] (+)= a[n] * dcls; token 0 is written plainly, tokens 1.. are read-modify-written (the patch kernels wrote
// them first).  16-byte accesses.  Strip 0 also writes dq0 = scale * (sum a*da*K - dot * sum a*K).
template <typename T>
__global__ __launch_bounds__(256) void cls_bwd_b_kernel(const T* __restrict__ qkv, const T* __restrict__ dcls,
                                                        const float* __restrict__ prob_g,
                                                        const float* __restrict__ da_g, const float* __restrict__ parts,
                                                        T* __restrict__ dqkv, int N, int heads) {
    const int bh = blockIdx.y, b = bh / heads, hh = bh % heads, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const float scale = rsqrtf((float)HD);
    const float* p = parts + (int64_t)bh * NS * BW;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) dot += p[k * BW];
    if (blockIdx.x == 0 && threadIdx.x < HD) {
        float u = 0.f, wv = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) { u += p[k * BW + 1 + threadIdx.x]; wv += p[k * BW + 1 + HD + threadIdx.x]; }
        // q0 was pre-multiplied by scale in phase A: the logits' derivative w.r.t. q0 carries one factor of scale
        st<T>(dqkv + (int64_t)b * N * tok + hh * HD + threadIdx.x, scale * (u - dot * wv));
    }
    for (int it = threadIdx.x; it < 64 * 16 * 2; it += 256) {
        const int which = it / (64 * 16), rem = it % (64 * 16), n = blockIdx.x * 64 + rem / 16, c = (rem % 16) * 4;
        if (n >= N) continue;
        const float a = prob_g[(int64_t)bh * N + n];
        const float w = which == 0 ? scale * a * (da_g[(int64_t)bh * N + n] - dot) : a;
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

bool focus_traj_cls_ok(int N, int d) { return d == HD && N >= 1; }

// floats of scratch the cls kernels need (forward and backward use the same bound)
size_t focus_traj_cls_scratch_floats(int B, int N, int heads) {
    return (size_t)2 * B * heads * N + (size_t)B * heads * NS * (BW > FW ? BW : FW) + 64;
}

int focus_traj_cls_fwd(const void* qkv, void* cls_out, float* cls_lse, float* scratch, int B, int N, int heads, int dtype,
                       hipStream_t s) {
    dim3 g(NS, B * heads);
    if (dtype == FOCUS_BF16) {
        hipLaunchKernelGGL((cls_fwd_part_kernel<bf16_t>), g, dim3(256), 0, s, (const bf16_t*)qkv, scratch, N, heads);
        hipLaunchKernelGGL((cls_fwd_comb_kernel<bf16_t>), dim3(B * heads), dim3(64), 0, s, scratch, (bf16_t*)cls_out, cls_lse, heads);
    } else {
        hipLaunchKernelGGL((cls_fwd_part_kernel<float>), g, dim3(256), 0, s, (const float*)qkv, scratch, N, heads);
        hipLaunchKernelGGL((cls_fwd_comb_kernel<float>), dim3(B * heads), dim3(64), 0, s, scratch, (float*)cls_out, cls_lse, heads);
    }
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

int focus_traj_cls_bwd(const void* qkv, const float* cls_lse, const void* dcls, void* dqkv, float* scratch, int B,
                       int N, int heads, int dtype, hipStream_t s) {
    // scratch: focus_traj_cls_scratch_floats(): prob | da | per-split partial sums
    float* prob_g = scratch;
    float* da_g = scratch + (size_t)B * heads * N;
    float* parts = da_g + (size_t)B * heads * N;
    dim3 ga(NS, B * heads), gb((N + 63) / 64, B * heads);
    if (dtype == FOCUS_BF16) {
        hipLaunchKernelGGL((cls_bwd_a_kernel<bf16_t>), ga, dim3(256), 0, s, (const bf16_t*)qkv, cls_lse,
                           (const bf16_t*)dcls, prob_g, da_g, parts, N, heads);
        hipLaunchKernelGGL((cls_bwd_b_kernel<bf16_t>), gb, dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)dcls,
                           prob_g, da_g, parts, (bf16_t*)dqkv, N, heads);
    } else {
        hipLaunchKernelGGL((cls_bwd_a_kernel<float>), ga, dim3(256), 0, s, (const float*)qkv, cls_lse,
                           (const float*)dcls, prob_g, da_g, parts, N, heads);
        hipLaunchKernelGGL((cls_bwd_b_kernel<float>), gb, dim3(256), 0, s, (const float*)qkv, (const float*)dcls, prob_g,
                           da_g, parts, (float*)dqkv, N, heads);
    }
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
