// slot_attn.hip -- STEVE slot-attention inverted softmax + weighted mean (steve.py:76-83), fwd and bwd.
// One launch streams k_t and v_t exactly once: per input row n the wave computes the K slot logits
// (K wave reductions over D), the softmax over slots, writes attn_vis and accumulates the un-normalised
// update sum_n (attn+eps)[n,k] * v[n,:] in registers (K x D/64 per lane).  A finish kernel reduces the
// per-chunk partials and applies the 1/sum_n normalisation.  HBM-bound: 2*N*D*esize read per (b,t,iter).
#include "focus_common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;

template <typename T, int KP, int DV>
__global__ __launch_bounds__(256) void slot_fwd_kernel(const T* __restrict__ kt, const T* __restrict__ vt,
                                                       int64_t kv_bs, const T* __restrict__ q,
                                                       T* __restrict__ attn, int64_t attn_bs,
                                                       float* __restrict__ partial, int N, int K, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // q [K][D], then reduce area
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < K * D; i += 256) sm[i] = ld<T>(q + (int64_t)b * K * D + i);
    __syncthreads();
    float U[KP][DV], cs[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        cs[k] = 0.f;
#pragma unroll
        for (int j = 0; j < DV; ++j) U[k][j] = 0.f;
    }
    const int n_end = min(N, (chunk + 1) * ROWS_PER_BLOCK);
    for (int n = chunk * ROWS_PER_BLOCK + w; n < n_end; n += 4) {
        const T* kr = kt + (int64_t)b * kv_bs + (int64_t)n * D;
        const T* vr = vt + (int64_t)b * kv_bs + (int64_t)n * D;
        float kv[DV], vv[DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int e = j * 64 + lane;
            kv[j] = e < D ? ld<T>(kr + e) : 0.f;
            vv[j] = e < D ? ld<T>(vr + e) : 0.f;
        }
        float lg[KP];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            float p = 0.f;
            if (k < K) {
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) p += kv[j] * sm[k * D + e];
                }
            }
            lg[k] = k < K ? wave_sum(p) : -INFINITY;
            m = fmaxf(m, lg[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) { lg[k] = k < K ? __expf(lg[k] - m) : 0.f; den += lg[k]; }
        const float inv = 1.f / den;
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float a = lg[k] * inv;
            if (k == lane) mine = a;
            if (k < K) {
                const float ae = a + eps;
                cs[k] += ae;
#pragma unroll
                for (int j = 0; j < DV; ++j) U[k][j] += ae * vv[j];
            }
        }
        if (lane < K) st<T>(attn + (int64_t)b * attn_bs + (int64_t)n * K + lane, mine);
    }
    // combine the 4 waves, write partial[b][chunk][k][0..D-1], colsum at [..][D]
    float* red = sm + K * D;  // [4][64]
    float* out = partial + ((int64_t)b * nchunk + chunk) * K * (D + 1);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K)
#pragma unroll
        for (int j = 0; j < DV + 1; ++j) {
            const float val = j < DV ? U[k][j < DV ? j : 0] : cs[k];
            __syncthreads();
            red[w * 64 + lane] = val;
            __syncthreads();
            if (w == 0) {
                const float t = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
                if (j < DV) { const int e = j * 64 + lane; if (e < D) out[k * (D + 1) + e] = t; }
                else if (lane == 0) out[k * (D + 1) + D] = t;  // every lane of a wave holds that wave's row sum
            }
        }
    }
}

// grid (K, B): reduce partials over chunks, normalise.
template <typename T>
__global__ void slot_fwd_finish(const float* __restrict__ partial, T* __restrict__ upd, float* __restrict__ colsum,
                                int nchunk, int K, int D) {
    const int k = blockIdx.x, b = blockIdx.y;
    const float* p = partial + (int64_t)b * nchunk * K * (D + 1) + (int64_t)k * (D + 1);
    float c = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) c += p[(int64_t)ch * K * (D + 1) + D];
    if (threadIdx.x == 0) colsum[b * K + k] = c;
    for (int e = threadIdx.x; e < D; e += blockDim.x) {
        float s = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) s += p[(int64_t)ch * K * (D + 1) + e];
        st<T>(upd + ((int64_t)b * K + k) * D + e, s / c);
    }
}

template <typename T, int KP, int DV>
__global__ __launch_bounds__(256) void slot_bwd_kernel(const T* __restrict__ kt, const T* __restrict__ vt,
                                                       int64_t kv_bs, const T* __restrict__ q,
                                                       const T* __restrict__ attn, int64_t attn_bs,
                                                       const float* __restrict__ colsum, const T* __restrict__ upd,
                                                       const T* __restrict__ dupd, const T* __restrict__ dattn,
                                                       T* __restrict__ dkt, T* __restrict__ dvt, int accumulate,
                                                       float* __restrict__ partial, int N, int K, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // q [K][D] | dupd [K][D] | r[K] | cs[K] | red
    float* sq = sm;
    float* sdu = sm + K * D;
    float* sr = sdu + K * D;
    float* scs = sr + 32;
    float* red = scs + 32;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < K * D; i += 256) {
        sq[i] = ld<T>(q + (int64_t)b * K * D + i);
        sdu[i] = ld<T>(dupd + (int64_t)b * K * D + i);
    }
    __syncthreads();
    // r[k] = dupd[k,:].upd[k,:]  (one wave per slot, round robin)
    for (int k = w; k < K; k += 4) {
        float p = 0.f;
        for (int e = lane; e < D; e += 64) p += sdu[k * D + e] * ld<T>(upd + ((int64_t)b * K + k) * D + e);
        p = wave_sum(p);
        if (lane == 0) { sr[k] = p; scs[k] = colsum[b * K + k]; }
    }
    __syncthreads();
    float dQ[KP][DV];
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
        for (int j = 0; j < DV; ++j) dQ[k][j] = 0.f;
    const int n_end = min(N, (chunk + 1) * ROWS_PER_BLOCK);
    for (int n = chunk * ROWS_PER_BLOCK + w; n < n_end; n += 4) {
        const int64_t ro = (int64_t)b * kv_bs + (int64_t)n * D;
        float kv[DV], vv[DV], dv[DV], dk[DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int e = j * 64 + lane;
            kv[j] = e < D ? ld<T>(kt + ro + e) : 0.f;
            vv[j] = e < D ? ld<T>(vt + ro + e) : 0.f;
            dv[j] = 0.f; dk[j] = 0.f;
        }
        float av[KP], dav[KP];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            av[k] = 0.f; dav[k] = 0.f;
            if (k < K) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) p += sdu[k * D + e] * vv[j];
                }
                const float dw = wave_sum(p);
                av[k] = ld<T>(attn + (int64_t)b * attn_bs + (int64_t)n * K + k);
                const float wgt = (av[k] + eps) / scs[k];
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) dv[j] += wgt * sdu[k * D + e];
                }
                dav[k] = (dw - sr[k]) / scs[k];
                if (dattn) dav[k] += ld<T>(dattn + (int64_t)b * attn_bs + (int64_t)n * K + k);
                dot += av[k] * dav[k];
            }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (k < K) {
                const float dl = av[k] * (dav[k] - dot);
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) { dk[j] += dl * sq[k * D + e]; dQ[k][j] += dl * kv[j]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int e = j * 64 + lane;
            if (e < D) {
                float a = dk[j], c = dv[j];
                if (accumulate) { a += ld<T>(dkt + ro + e); c += ld<T>(dvt + ro + e); }
                st<T>(dkt + ro + e, a);
                st<T>(dvt + ro + e, c);
            }
        }
    }
    float* out = partial + ((int64_t)b * nchunk + chunk) * K * D;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K)
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            __syncthreads();
            red[w * 64 + lane] = dQ[k][j];
            __syncthreads();
            if (w == 0) {
                const int e = j * 64 + lane;
                if (e < D) out[k * D + e] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
            }
        }
    }
}

template <typename T>
__global__ void slot_bwd_finish(const float* __restrict__ partial, T* __restrict__ dq, int nchunk, int K, int D) {
    const int k = blockIdx.x, b = blockIdx.y;
    for (int e = threadIdx.x; e < D; e += blockDim.x) {
        float s = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) s += partial[(((int64_t)b * nchunk + ch) * K + k) * D + e];
        st<T>(dq + ((int64_t)b * K + k) * D + e, s);
    }
}

inline int nchunks(int N) { return (N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }

}  // namespace

extern "C" size_t focus_slot_attn_workspace_bytes(int B, int N, int K, int D) {
    return (size_t)B * nchunks(N) * K * (D + 1) * sizeof(float);
}

#define SLOT_DISPATCH(KERNEL_CALL)                                                          \
    do {                                                                                    \
        const int dv = (D + 63) / 64;                                                       \
        const int kp = K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : 32;                         \
        bool done = false;                                                                  \
        SLOT_CASE(4, 1) SLOT_CASE(4, 2) SLOT_CASE(4, 3) SLOT_CASE(4, 4)                     \
        SLOT_CASE(8, 1) SLOT_CASE(8, 2) SLOT_CASE(8, 3) SLOT_CASE(8, 4)                     \
        SLOT_CASE(16, 1) SLOT_CASE(16, 2) SLOT_CASE(16, 3) SLOT_CASE(16, 4)                 \
        SLOT_CASE(32, 1) SLOT_CASE(32, 2) SLOT_CASE(32, 3) SLOT_CASE(32, 4)                 \
        if (!done) return FOCUS_ERR_SHAPE;                                                  \
    } while (0)

extern "C" int focus_slot_attn_fwd(const void* k_t, const void* v_t, int64_t kv_bs, const void* q, void* attn_vis,
                                   int64_t attn_bs, void* upd, float* colsum, void* partial, size_t partial_bytes,
                                   int B, int N, int K, int D, float eps, int dtype, void* stream) {
    if (!k_t || !v_t || !q || !attn_vis || !upd || !colsum || !partial) return FOCUS_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || K > 32 || D <= 0 || D > 256 || B > 65535) return FOCUS_ERR_SHAPE;
    if (partial_bytes < focus_slot_attn_workspace_bytes(B, N, K, D)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nchunks(N), B);
    const size_t lds = ((size_t)K * D + 256) * sizeof(float);
#define SLOT_CASE(KP_, DV_)                                                                                   \
    if (!done && kp == KP_ && dv == DV_) {                                                                    \
        done = true;                                                                                          \
        if (dtype == FOCUS_BF16)                                                                              \
            hipLaunchKernelGGL((slot_fwd_kernel<bf16_t, KP_, DV_>), grid, dim3(256), lds, s, (const bf16_t*)k_t, \
                               (const bf16_t*)v_t, kv_bs, (const bf16_t*)q, (bf16_t*)attn_vis, attn_bs,       \
                               (float*)partial, N, K, D, eps);                                                \
        else                                                                                                  \
            hipLaunchKernelGGL((slot_fwd_kernel<float, KP_, DV_>), grid, dim3(256), lds, s, (const float*)k_t, \
                               (const float*)v_t, kv_bs, (const float*)q, (float*)attn_vis, attn_bs,          \
                               (float*)partial, N, K, D, eps);                                                \
    }
    SLOT_DISPATCH();
#undef SLOT_CASE
    FOCUS_CHECK_LAUNCH();
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((slot_fwd_finish<bf16_t>), dim3(K, B), dim3(64), 0, s, (const float*)partial, (bf16_t*)upd,
                           colsum, nchunks(N), K, D);
    else
        hipLaunchKernelGGL((slot_fwd_finish<float>), dim3(K, B), dim3(64), 0, s, (const float*)partial, (float*)upd,
                           colsum, nchunks(N), K, D);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_slot_attn_bwd(const void* k_t, const void* v_t, int64_t kv_bs, const void* q,
                                   const void* attn_vis, int64_t attn_bs, const float* colsum, const void* upd,
                                   const void* dupd, const void* dattn_vis, void* dk_t, void* dv_t, int accumulate,
                                   void* dq, void* partial, size_t partial_bytes, int B, int N, int K, int D,
                                   float eps, int dtype, void* stream) {
    if (!k_t || !v_t || !q || !attn_vis || !colsum || !upd || !dupd || !dk_t || !dv_t || !dq || !partial)
        return FOCUS_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || K > 32 || D <= 0 || D > 256 || B > 65535) return FOCUS_ERR_SHAPE;
    if (partial_bytes < focus_slot_attn_workspace_bytes(B, N, K, D)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nchunks(N), B);
    const size_t lds = ((size_t)2 * K * D + 64 + 256) * sizeof(float);
#define SLOT_CASE(KP_, DV_)                                                                                    \
    if (!done && kp == KP_ && dv == DV_) {                                                                     \
        done = true;                                                                                           \
        if (dtype == FOCUS_BF16)                                                                               \
            hipLaunchKernelGGL((slot_bwd_kernel<bf16_t, KP_, DV_>), grid, dim3(256), lds, s, (const bf16_t*)k_t, \
                               (const bf16_t*)v_t, kv_bs, (const bf16_t*)q, (const bf16_t*)attn_vis, attn_bs,  \
                               colsum, (const bf16_t*)upd, (const bf16_t*)dupd, (const bf16_t*)dattn_vis,      \
                               (bf16_t*)dk_t, (bf16_t*)dv_t, accumulate, (float*)partial, N, K, D, eps);       \
        else                                                                                                   \
            hipLaunchKernelGGL((slot_bwd_kernel<float, KP_, DV_>), grid, dim3(256), lds, s, (const float*)k_t, \
                               (const float*)v_t, kv_bs, (const float*)q, (const float*)attn_vis, attn_bs,     \
                               colsum, (const float*)upd, (const float*)dupd, (const float*)dattn_vis,         \
                               (float*)dk_t, (float*)dv_t, accumulate, (float*)partial, N, K, D, eps);         \
    }
    SLOT_DISPATCH();
#undef SLOT_CASE
    FOCUS_CHECK_LAUNCH();
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((slot_bwd_finish<bf16_t>), dim3(K, B), dim3(64), 0, s, (const float*)partial, (bf16_t*)dq,
                           nchunks(N), K, D);
    else
        hipLaunchKernelGGL((slot_bwd_finish<float>), dim3(K, B), dim3(64), 0, s, (const float*)partial, (float*)dq,
                           nchunks(N), K, D);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
