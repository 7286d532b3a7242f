// small_attn.hip -- multi-head attention over a handful of tokens, one launch each way.
//
// STEVE's slot predictor (transformer.py:4-49 inside steve.py:98) attends over the K = 11 slots of a frame: 4 heads of
// d = 48, B = 32 clips -- 128 independent 11 x 11 problems of 0.1 MFLOP.  As strided batched GEMMs + a row softmax that is
// 3 launches forward and 5 backward per frame (generic-kernel GEMMs of 8-9 us each: the shapes fit no MFMA tile), 24
// frames per step.  Here one 256-thread workgroup owns one (clip, head): q, k, v [<= 16 tokens][<= 64 channels] sit in LDS
// as fp32 and every product is a few dozen FMAs per thread.  (One wave per problem was LDS-latency bound: 13 / 21 us
// forward / backward; the four waves of a workgroup share each loop.)
//   forward : S = scale q k^T, P = softmax_rows(S) (stored, bf16/fp32 like the inputs), out = P v
//   backward: dP = dout v^T, dV = P^T dout, dS = scale P (dP - rowsum(P dP)), dq = dS k, dk = dS^T q
// P is rounded to the storage type before it multiplies v, as the unfused path's stored probabilities were.
#include "focus_common.h"

namespace {

constexpr int TMAX = 16, DMAX = 64, NT = 256;           // tokens, head channels, threads of the workgroup that owns a problem

struct SmallAttnLds {
    float q[TMAX][DMAX + 1], k[TMAX][DMAX + 1], v[TMAX][DMAX + 1], o[TMAX][DMAX + 1];   // o: dout in the backward
    float s[TMAX][TMAX + 1], t[TMAX][TMAX + 1];
};

#define DISPATCH_T(dtype, EXPR_T)                                     \
    do {                                                              \
        if ((dtype) == FOCUS_BF16) { typedef bf16_t T; EXPR_T; }      \
        else { typedef float T; EXPR_T; }                             \
    } while (0)

template <typename T> __device__ __forceinline__ float stored(float x) { T t; st<T>(&t, x); return ld<T>(&t); }

template <typename T>
__device__ __forceinline__ void load_rows(float (*dst)[DMAX + 1], const T* src, int rows, int d, int64_t rs, int lane) {
    for (int e = lane; e < rows * d; e += NT) dst[e / d][e % d] = ld<T>(src + (int64_t)(e / d) * rs + e % d);
}

template <typename T>
__global__ __launch_bounds__(NT) void small_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                  const T* __restrict__ v, T* __restrict__ att,
                                                                  T* __restrict__ out, int nprob, int heads, int N, int M,
                                                                  int d, float scale, int64_t ldq, int64_t ldk, int64_t ldv) {
    __shared__ SmallAttnLds L;
    const int lane = threadIdx.x;                        // (index within the problem's workgroup)
    const int p = blockIdx.x;                            // problem = (clip b, head h)
    const int b = p / heads, h = p - b * heads, C = heads * d;
    const T* qb = q + ((int64_t)b * N) * ldq + h * d;
    const T* kb = k + ((int64_t)b * M) * ldk + h * d;
    const T* vb = v + ((int64_t)b * M) * ldv + h * d;
    load_rows<T>(L.q, qb, N, d, ldq, lane);
    load_rows<T>(L.k, kb, M, d, ldk, lane);
    load_rows<T>(L.v, vb, M, d, ldv, lane);
    __syncthreads();
    for (int e = lane; e < N * M; e += NT) {
        const int i = e / M, j = e - i * M;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < d; ++c) a = fmaf(L.q[i][c], L.k[j][c], a);
        L.s[i][j] = a * scale;
    }
    __syncthreads();
    if (lane < N) {                                      // one lane per row: M <= 16 terms
        float m = -INFINITY, sum = 0.f;
        for (int j = 0; j < M; ++j) m = fmaxf(m, L.s[lane][j]);
        for (int j = 0; j < M; ++j) { const float e = __expf(L.s[lane][j] - m); L.s[lane][j] = e; sum += e; }
        const float inv = 1.f / sum;
        T* ar = att + ((int64_t)p * N + lane) * M;
        for (int j = 0; j < M; ++j) {
            const float pr = stored<T>(L.s[lane][j] * inv);   // the stored (rounded) probability is what multiplies v
            st<T>(ar + j, pr);
            L.s[lane][j] = pr;
        }
    }
    __syncthreads();
    T* ob = out + ((int64_t)b * N) * C + h * d;
    for (int e = lane; e < N * d; e += NT) {
        const int i = e / d, c = e - i * d;
        float a = 0.f;
#pragma unroll 4
        for (int j = 0; j < M; ++j) a = fmaf(L.s[i][j], L.v[j][c], a);
        st<T>(ob + (int64_t)i * C + c, a);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void small_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                  const T* __restrict__ v, const T* __restrict__ att,
                                                                  const T* __restrict__ dout, T* __restrict__ dq,
                                                                  T* __restrict__ dk, T* __restrict__ dv, int nprob,
                                                                  int heads, int N, int M, int d, float scale, int64_t ldq,
                                                                  int64_t ldk, int64_t ldv) {
    __shared__ SmallAttnLds L;
    const int lane = threadIdx.x;
    const int p = blockIdx.x;
    const int b = p / heads, h = p - b * heads, C = heads * d;
    // q / dq, k / dk, v / dv rows are ldq, ldk, ldv elements apart (the three may be column blocks of one [rows, 3C] matrix)
    const int64_t qo = ((int64_t)b * N) * ldq + h * d, ko = ((int64_t)b * M) * ldk + h * d, vo = ((int64_t)b * M) * ldv + h * d;
    const int64_t oo = ((int64_t)b * N) * C + h * d;
    load_rows<T>(L.q, q + qo, N, d, ldq, lane);
    load_rows<T>(L.k, k + ko, M, d, ldk, lane);
    load_rows<T>(L.v, v + vo, M, d, ldv, lane);
    load_rows<T>(L.o, dout + oo, N, d, C, lane);
    for (int e = lane; e < N * M; e += NT) L.s[e / M][e % M] = ld<T>(att + (int64_t)p * N * M + e);
    __syncthreads();
    // dV = P^T dout;  dP = dout v^T
    for (int e = lane; e < M * d; e += NT) {
        const int j = e / d, c = e - j * d;
        float a = 0.f;
#pragma unroll 4
        for (int i = 0; i < N; ++i) a = fmaf(L.s[i][j], L.o[i][c], a);
        st<T>(dv + vo + (int64_t)j * ldv + c, a);
    }
    for (int e = lane; e < N * M; e += NT) {
        const int i = e / M, j = e - i * M;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < d; ++c) a = fmaf(L.o[i][c], L.v[j][c], a);
        L.t[i][j] = a;
    }
    __syncthreads();
    if (lane < N) {                                      // dS = scale P (dP - sum_j P dP)
        float dot = 0.f;
        for (int j = 0; j < M; ++j) dot = fmaf(L.s[lane][j], L.t[lane][j], dot);
        for (int j = 0; j < M; ++j) L.t[lane][j] = scale * L.s[lane][j] * (L.t[lane][j] - dot);
    }
    __syncthreads();
    for (int e = lane; e < N * d; e += NT) {             // dq = dS k
        const int i = e / d, c = e - i * d;
        float a = 0.f;
#pragma unroll 4
        for (int j = 0; j < M; ++j) a = fmaf(L.t[i][j], L.k[j][c], a);
        st<T>(dq + qo + (int64_t)i * ldq + c, a);
    }
    for (int e = lane; e < M * d; e += NT) {             // dk = dS^T q
        const int j = e / d, c = e - j * d;
        float a = 0.f;
#pragma unroll 4
        for (int i = 0; i < N; ++i) a = fmaf(L.t[i][j], L.q[i][c], a);
        st<T>(dk + ko + (int64_t)j * ldk + c, a);
    }
}

bool small_attn_shape_ok(int B, int heads, int N, int M, int d) {
    return B > 0 && heads > 0 && N > 0 && M > 0 && d > 0 && N <= TMAX && M <= TMAX && d <= DMAX;
}

}  // namespace

extern "C" int focus_small_attn_ok(int N, int M, int d) { return N > 0 && M > 0 && d > 0 && N <= TMAX && M <= TMAX && d <= DMAX; }

extern "C" int focus_small_attn_fwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, void* att,
                                    void* out, int B, int heads, int N, int M, int d, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !att || !out) return FOCUS_ERR_NULL;
    if (!small_attn_shape_ok(B, heads, N, M, d) || ldq < heads * d || ldk < heads * d || ldv < heads * d) return FOCUS_ERR_SHAPE;
    const int nprob = B * heads;
    DISPATCH_T(dtype, hipLaunchKernelGGL((small_attn_fwd_kernel<T>), dim3(nprob), dim3(NT), 0,
                                         (hipStream_t)stream, (const T*)q, (const T*)k, (const T*)v, (T*)att, (T*)out, nprob,
                                         heads, N, M, d, scale, ldq, ldk, ldv));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_small_attn_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv,
                                    const void* att, const void* dout, void* dq, void* dk, void* dv, int B, int heads, int N,
                                    int M, int d, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !att || !dout || !dq || !dk || !dv) return FOCUS_ERR_NULL;
    if (!small_attn_shape_ok(B, heads, N, M, d) || ldq < heads * d || ldk < heads * d || ldv < heads * d) return FOCUS_ERR_SHAPE;
    const int nprob = B * heads;
    DISPATCH_T(dtype, hipLaunchKernelGGL((small_attn_bwd_kernel<T>), dim3(nprob), dim3(NT), 0,
                                         (hipStream_t)stream, (const T*)q, (const T*)k, (const T*)v, (const T*)att,
                                         (const T*)dout, (T*)dq, (T*)dk, (T*)dv, nprob, heads, N, M, d, scale, ldq, ldk, ldv));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
