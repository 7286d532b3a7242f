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

// sum over the 8 lanes of a key group (lanes 8k .. 8k+7)
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v + __shfl_xor(v, 4, 64);
}
// sum over the 8 key groups of a wave (same channel lane in every group)
__device__ __forceinline__ float across_groups_sum(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// The N keys of one (b,h) are split over NS workgroups (flash-decoding style): with one workgroup per (b,h) only 96
// of the 256 CUs had work and each streamed its 400 KB of K/V rows at latency (39 / 58 us per call).
constexpr int NS = 8;            // key splits per (b,h)
constexpr int FW = 2 + HD;       // forward partial: max, sum, out[64]
constexpr int BW = 1 + 2 * HD;   // backward partial: sum a*da, sum a*da*K[64], sum a*K[64]

// Forward, part: split sp of (b,h) -> scratch[bh][sp] = (m, sum exp(l - m), sum exp(l - m) * V[n,:])
// Eight lanes share a key: each reads 16 bytes (8 channels) of its K and V rows, so a wave instruction covers 8 whole
// 128-byte head rows.  (One thread per key read its row in sixteen 8-byte pieces, 4.6 KB apart from its neighbours'
// -- every piece a separate cache-line access -- and V in 2-byte pieces: 22.6 us per launch for 19 MB.)
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
    const int l8 = threadIdx.x & 7, kg = threadIdx.x >> 3, w = threadIdx.x >> 6;     // channel octet, key group 0..31
    float q8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q8[i] = sq[l8 * 8 + i];
    float m = -INFINITY, ssum = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c0 = n0; c0 < n1; c0 += 256) {
        // logits of keys c0 .. c0+255: 8 sub-passes of 32 keys
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const int n = c0 + s8 * 32 + kg;
            float d = 0.f;
            if (n < n1) {
                float k8[8];
                ld8<T>(base + (int64_t)n * tok + C + l8 * 8, k8);
#pragma unroll
                for (int i = 0; i < 8; ++i) d = fmaf(k8[i], q8[i], d);
            }
            d = group8_sum(d);
            if (l8 == 0) prob[s8 * 32 + kg] = n < n1 ? d : -INFINITY;
        }
        __syncthreads();
        const float l = prob[threadIdx.x];
        const float mc = fmaxf(m, block_max(l, red));
        __syncthreads();
        const float e = l > -INFINITY ? __expf(l - mc) : 0.f;
        prob[threadIdx.x] = e;
        const float resc = __expf(m - mc);                    // 0 on the first pass (m = -inf)
        ssum = ssum * resc + block_sum(e, red);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= resc;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const int n = c0 + s8 * 32 + kg;
            if (n < n1) {
                float v8[8];
                ld8<T>(base + (int64_t)n * tok + 2 * C + l8 * 8, v8);
                const float pj = prob[s8 * 32 + kg];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(pj, v8[i], acc[i]);
            }
        }
        m = mc;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float t = across_groups_sum(acc[i]);
        if ((threadIdx.x & 63) < 8) part[w][l8 * 8 + i] = t;
    }
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
// Eight lanes share a key (16-byte K and V pieces, as in the forward); the K octet that gave the logit stays in
// registers for the two weighted sums, so K is read once.
template <typename T>
__global__ __launch_bounds__(256) void cls_bwd_a_kernel(const T* __restrict__ qkv, const float* __restrict__ cls_lse,
                                                        const T* __restrict__ dcls, float* __restrict__ prob_g,
                                                        float* __restrict__ da_g, float* __restrict__ parts, int N,
                                                        int heads) {
    __shared__ float sq[HD], sd[HD], red[4], part[2][4][HD];
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
    const int l8 = threadIdx.x & 7, kg = threadIdx.x >> 3, w = threadIdx.x >> 6;
    float q8[8], d8[8], u[8], wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { q8[i] = sq[l8 * 8 + i]; d8[i] = sd[l8 * 8 + i]; u[i] = 0.f; wv[i] = 0.f; }
    float dot = 0.f;
    for (int n = n0 + kg; n < n0 + ((n1 - n0 + 31) / 32) * 32; n += 32) {     // (uniform trip count: the shuffles need every lane)
        float k8[8], a = 0.f, da = 0.f;
        const bool ok = n < n1;
        if (ok) {
            float v8[8];
            const T* row = base + (int64_t)n * tok;
            ld8<T>(row + C + l8 * 8, k8);
            ld8<T>(row + 2 * C + l8 * 8, v8);
#pragma unroll
            for (int i = 0; i < 8; ++i) { a = fmaf(k8[i], q8[i], a); da = fmaf(v8[i], d8[i], da); }
        }
        a = group8_sum(a);
        da = group8_sum(da);
        if (ok) {
            a = __expf(a - lse);
            if (l8 == 0) { prob_g[(int64_t)bh * N + n] = a; da_g[(int64_t)bh * N + n] = da; dot += a * da; }
            const float ad = a * da;
#pragma unroll
            for (int i = 0; i < 8; ++i) { u[i] = fmaf(ad, k8[i], u[i]); wv[i] = fmaf(a, k8[i], wv[i]); }
        }
    }
    dot = block_sum(dot, red);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float tu = across_groups_sum(u[i]), tw = across_groups_sum(wv[i]);
        if ((threadIdx.x & 63) < 8) { part[0][w][l8 * 8 + i] = tu; part[1][w][l8 * 8 + i] = tw; }
    }
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
// them first).  16-byte pieces.  Strip 0 also writes dq0 = scale * (sum a*da*K - dot * sum a*K).
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
    for (int it = threadIdx.x; it < 64 * 8 * 2; it += 256) {     // 16-byte pieces: (part, row, channel octet)
        const int which = it / (64 * 8), rem = it % (64 * 8), n = blockIdx.x * 64 + rem / 8, c = (rem % 8) * 8;
        if (n >= N) continue;
        const float a = prob_g[(int64_t)bh * N + n];
        const float w = which == 0 ? scale * a * (da_g[(int64_t)bh * N + n] - dot) : a;
        const T* src = which == 0 ? qkv + (int64_t)b * N * tok + hh * HD + c          // q0 (token 0, q part)
                                  : dcls + (int64_t)b * C + hh * HD + c;
        float v[8], o[8];
        ld8<T>(src, v);
        T* dst = dqkv + ((int64_t)b * N + n) * tok + (which == 0 ? C : 2 * C) + hh * HD + c;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = w * v[i];
        if (n > 0) {
            float old[8];
            ld8<T>(dst, old);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += old[i];
        }
        st8<T>(dst, o);
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
