// gumbel.hip -- the row passes over the [frames * tokens, vocabulary] tensors of STEVE.forward, one workgroup per row with
// the row held in registers:
//   * focus_gumbel_fwd / _bwd: log_softmax -> Gumbel-softmax relaxation (straight-through one-hot when `hard`) -> the
//     arg-max of a second, independent Gumbel perturbation (the decoder's target tokens)      steve.py:262-271, utils.py:47-61
//   * focus_xent_rows_fwd / _bwd: label-smoothing cross entropy with fp32 or bf16 logits       steve.py:303-306, losses.py:53-59
// As ATen launches these are 15 elementwise / reduction passes over a tensor of 3.2 GB (batch 8) with as many temporaries;
// here the forward reads the logits once and writes the sample once, the backward reads logits and d(sample) and writes
// d(logits).  HBM-bound: 2 and 3 row passes.
//
// The Exp(1) draws behind the Gumbel noise are either read (parity tests pass the reference's own draws) or generated in
// the kernel from a device-side seed -- a hash of (seed, row, column), so the backward rebuilds the forward's noise instead
// of reading 3.2 GB of it back.  Generated noise uses the hardware log2 / exp2; supplied noise the accurate ones.
#include "focus_common.h"

namespace {

constexpr float kTiny = 1.17549435e-38f;        // torch.finfo(float32).tiny
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// E ~ Exp(1) from 23 hashed bits: u = (k + 1/2) / 2^23 in (0, 1), E = -ln u.  (tests/test_gpu_gumbel.py holds the numpy twin.)
// row_key = mix32(row ^ seed[0]) (+ the row's high word), s1 = seed[1]: two hash rounds per draw, one of them per row.
__device__ __forceinline__ float exp1_draw(uint32_t s1, uint32_t row_key, uint32_t col) {
    const uint32_t h = mix32(row_key ^ (col * 0x9E3779B9u) ^ s1);
    const float u = ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return -kLn2 * __builtin_amdgcn_logf(u);
}
template <bool GEN> __device__ __forceinline__ float ln(float x) { return GEN ? kLn2 * __builtin_amdgcn_logf(x) : logf(x); }
template <bool GEN> __device__ __forceinline__ float ex(float x) { return GEN ? __builtin_amdgcn_exp2f(x * kLog2e) : expf(x); }
// a / b as the reference divides, or a * (1 / b) with the reciprocal formed once per row (generated-noise path)
template <bool GEN> __device__ __forceinline__ float dv(float a, float b, float rb) { return GEN ? a * rb : a / b; }

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
// arg-max over the workgroup, the lowest index among equal maxima
__device__ __forceinline__ Best block_argmax(Best b, float* redv, int* redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best other = {__shfl_xor(b.v, o, 64), __shfl_xor(b.i, o, 64)};
        b = better(b, other);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { redv[w] = b.v; redi[w] = b.i; }
    __syncthreads();
    Best t = {redv[0], redi[0]};
#pragma unroll
    for (int i = 1; i < 4; ++i) t = better(t, Best{redv[i], redi[i]});
    return t;
}

// Row layout: 256 threads, thread t owns the 8-element chunks at columns (c * 256 + t) * 8, c < NC  (V <= 2048 * NC).
template <typename T, int NC>
__device__ __forceinline__ void load_row(const T* row, int V, float (&v)[NC][8], float fill) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (c * 256 + (int)threadIdx.x) * 8;
        if (col < V) ld8<T>(row + col, v[c]);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = fill;
        }
    }
}
template <typename T, int NC>
__device__ __forceinline__ void store_row(T* row, int V, const float (&v)[NC][8]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (c * 256 + (int)threadIdx.x) * 8;
        if (col < V) st8<T>(row + col, v[c]);
    }
}
// G = -ln(E + tiny) for the chunk at `col` of row r
template <bool GEN, int NC>
__device__ __forceinline__ void gumbel_row(const float* e, uint32_t s0, uint32_t s1, int64_t r, int V, float (&g)[NC][8]) {
    const uint32_t row_key = GEN ? mix32((uint32_t)r ^ s0) + (uint32_t)(r >> 32) : 0u;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (c * 256 + (int)threadIdx.x) * 8;
        float ev[8];
        if (!GEN) {
            if (col < V) ld8<float>(e + r * V + col, ev);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ev[i] = 1.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float E = GEN ? exp1_draw(s1, row_key, (uint32_t)(col + i)) : ev[i];
            g[c][i] = -ln<GEN>(E + kTiny);
        }
    }
}

template <typename T, typename TZ, int NC, bool GEN>
__global__ __launch_bounds__(256) void gumbel_fwd_kernel(const T* __restrict__ x, const float* __restrict__ e_soft,
                                                         const float* __restrict__ e_hard, const uint32_t* __restrict__ seed,
                                                         TZ* __restrict__ z, int64_t* __restrict__ target,
                                                         float* __restrict__ stats, int V, float tau, int hard) {
    __shared__ float red[16];
    __shared__ int redi[8];
    const int64_t r = blockIdx.x;
    float v[NC][8], g[NC][8];
    load_row<T, NC>(x + r * V, V, v, -INFINITY);
    // log_softmax (ATen's form: (x - max) - log(sum))
    float m1 = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) m1 = fmaxf(m1, v[c][i]);
    m1 = block_max(m1, red);
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) s1 += ex<GEN>(v[c][i] - m1);
    s1 = block_sum(s1, red);
    const float ls1 = ln<GEN>(s1);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] = (v[c][i] - m1) - ls1;
    uint32_t sd[4] = {0, 0, 0, 0};
    if (GEN) { sd[0] = seed[0]; sd[1] = seed[1]; sd[2] = seed[2]; sd[3] = seed[3]; }
    // the hard sample, read only through its arg-max: arg-max of the perturbed log-probabilities (softmax is monotonic)
    if (target) {
        gumbel_row<GEN, NC>(e_hard, sd[2], sd[3], r, V, g);
        Best b = {-INFINITY, 0x7fffffff};
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) b = better(b, Best{v[c][i] + g[c][i], (c * 256 + (int)threadIdx.x) * 8 + i});
        b = block_argmax(b, red, redi);
        if (threadIdx.x == 0) target[r] = b.i;
    }
    // the relaxed sample: softmax((logp + G) / tau)
    gumbel_row<GEN, NC>(e_soft, sd[0], sd[1], r, V, g);
    float m2 = -INFINITY;
    const float rtau = 1.f / tau;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[c][i] = dv<GEN>(v[c][i] + g[c][i], tau, rtau);
            m2 = fmaxf(m2, v[c][i]);
        }
    m2 = block_max(m2, red);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[c][i] = ex<GEN>(v[c][i] - m2);
            s2 += v[c][i];
        }
    s2 = block_sum(s2, red);
    const float rs2 = 1.f / s2;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] = dv<GEN>(v[c][i], s2, rs2);
    if (hard) {                                   // one_hot - soft.detach() + soft, in that order (utils.py:58-60)
        Best b = {-INFINITY, 0x7fffffff};
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) b = better(b, Best{v[c][i], (c * 256 + (int)threadIdx.x) * 8 + i});
        b = block_argmax(b, red, redi);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float one = ((c * 256 + (int)threadIdx.x) * 8 + i) == b.i ? 1.f : 0.f;
                v[c][i] = (one - v[c][i]) + v[c][i];
            }
    }
    store_row<TZ, NC>(z + r * V, V, v);
    if (threadIdx.x == 0) *reinterpret_cast<float4*>(stats + r * 4) = make_float4(m1, ls1, m2, s2);
}

template <typename T, typename TZ, int NC, bool GEN>
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(const T* __restrict__ x, const float* __restrict__ e_soft,
                                                         const uint32_t* __restrict__ seed, const float* __restrict__ stats,
                                                         const TZ* __restrict__ dz, T* __restrict__ dx, int V, float tau) {
    __shared__ float red[16];
    const int64_t r = blockIdx.x;
    float lp[NC][8], so[NC][8], d[NC][8];
    const float4 st = *reinterpret_cast<const float4*>(stats + r * 4);
    load_row<T, NC>(x + r * V, V, lp, -INFINITY);
    load_row<TZ, NC>(dz + r * V, V, d, 0.f);
    uint32_t s0 = 0, s1 = 0;
    if (GEN) { s0 = seed[0]; s1 = seed[1]; }
    gumbel_row<GEN, NC>(e_soft, s0, s1, r, V, so);
    float t = 0.f;
    const float rtau = 1.f / tau, rs2 = 1.f / st.w;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            lp[c][i] = (lp[c][i] - st.x) - st.y;                                  // log-probability
            so[c][i] = dv<GEN>(ex<GEN>(dv<GEN>(lp[c][i] + so[c][i], tau, rtau) - st.z), st.w, rs2);   // the forward's soft sample
            t += d[c][i] * so[c][i];
        }
    t = block_sum(t, red);
    float sd = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            d[c][i] = dv<GEN>((d[c][i] - t) * so[c][i], tau, rtau);                           // softmax', then the division by tau
            sd += d[c][i];
        }
    sd = block_sum(sd, red);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) d[c][i] = d[c][i] - ex<GEN>(lp[c][i]) * sd;   // log_softmax'
    store_row<T, NC>(dx + r * V, V, d);
}

// ---- label smoothing cross entropy over rows of fp32 / bf16 logits ------------------------------------------------------
template <typename T, int NC>
__global__ __launch_bounds__(256) void xent_rows_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                            float* __restrict__ loss_rows, float* __restrict__ lse_out, int V,
                                                            float smoothing) {
    __shared__ float red[16];
    const int64_t r = blockIdx.x;
    float v[NC][8];
    load_row<T, NC>(logits + r * V, V, v, -INFINITY);
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, v[c][i]);
    m = block_max(m, red);
    float s = 0.f, sx = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s += expf(v[c][i] - m);
            sx += (c * 256 + (int)threadIdx.x) * 8 + i < V ? v[c][i] : 0.f;
        }
    s = block_sum(s, red);
    sx = block_sum(sx, red);
    if (threadIdx.x == 0) {
        const float lse = m + logf(s);
        const float xt = ld<T>(logits + r * V + target[r]);
        loss_rows[r] = (1.f - smoothing) * (lse - xt) + smoothing * (lse - sx / (float)V);
        lse_out[r] = lse;
    }
}
// dlogits = ((softmax - (1 - smoothing) onehot - smoothing / V) / R) * g,   g = d(loss) / d(mean of the rows), on the device
template <typename T, int NC>
__global__ __launch_bounds__(256) void xent_rows_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                            const float* __restrict__ lse_in, const float* __restrict__ g,
                                                            T* __restrict__ dlogits, int V, float smoothing, float rows) {
    const int64_t r = blockIdx.x;
    float v[NC][8];
    load_row<T, NC>(logits + r * V, V, v, -INFINITY);
    const float lse = lse_in[r], gs = g[0], conf = 1.f - smoothing, sm = smoothing / (float)V;
    const int t = (int)target[r];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = (c * 256 + (int)threadIdx.x) * 8 + i;
            v[c][i] = ((expf(v[c][i] - lse) - (col == t ? conf : 0.f) - sm) / rows) * gs;
        }
    store_row<T, NC>(dlogits + r * V, V, v);
}

// ---- out = residual + dropout(y): the three residual branches of a decoder block (transformer.py:45-47, :147-163) -----------
// keep(i) = 16 hashed bits of (seed, i) >= thr, thr = round(p * 65536); survivors are scaled by 65536 / (65536 - thr).
// One 32-bit hash serves two neighbouring elements.  The backward of the branch is the same pass over the incoming gradient
// with res == NULL: the mask is a function of (seed, index) and is never stored.   (numpy twin: tests/test_gpu_gumbel.py)
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                          const uint32_t* __restrict__ seed, uint32_t thr, T* __restrict__ out,
                                                          int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const uint32_t s0 = seed[0], s1 = seed[1];
    const float inv = 65536.f / (float)(65536u - thr);
    float v[8], r[8];
    ld8<T>(y + i, v);
    if (res) ld8<T>(res + i, r);
    const uint64_t pair = (uint64_t)i >> 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t pj = pair + j;
        const uint32_t h = mix32((uint32_t)pj ^ s0) ^ ((uint32_t)(pj >> 32) * 0x9E3779B9u + s1);
        const float a = (h & 0xffffu) >= thr ? v[2 * j] * inv : 0.f;
        const float b = (h >> 16) >= thr ? v[2 * j + 1] * inv : 0.f;
        v[2 * j] = res ? r[2 * j] + a : a;
        v[2 * j + 1] = res ? r[2 * j + 1] + b : b;
    }
    st8<T>(out + i, v);
}

bool rows_ok(int64_t R, int V) { return R > 0 && R < (1ll << 31) && V >= 8 && V % 8 == 0 && V <= 8192; }

}  // namespace

#define GUMBEL_NC(V, ...)                                  \
    do {                                                   \
        if ((V) <= 2048) { constexpr int NC = 1; __VA_ARGS__; } \
        else if ((V) <= 4096) { constexpr int NC = 2; __VA_ARGS__; } \
        else { constexpr int NC = 4; __VA_ARGS__; }        \
    } while (0)
// the sample (and its gradient) in the logits' type, or in bf16 under fp32 logits (a bf16 decoder behind an fp32 encoder)
#define GUMBEL_TZ(dtype, dtype_z, ...)                     \
    do {                                                   \
        if ((dtype) == FOCUS_BF16) { typedef bf16_t T; typedef bf16_t TZ; __VA_ARGS__; } \
        else if ((dtype_z) == FOCUS_BF16) { typedef float T; typedef bf16_t TZ; __VA_ARGS__; } \
        else { typedef float T; typedef float TZ; __VA_ARGS__; } \
    } while (0)
#define GUMBEL_T(dtype, ...)                               \
    do {                                                   \
        if ((dtype) == FOCUS_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else { typedef float T; __VA_ARGS__; }             \
    } while (0)

extern "C" int focus_rows_ok(int64_t R, int V, int dtype) {
    return rows_ok(R, V) && (dtype == FOCUS_F32 || dtype == FOCUS_BF16) ? 1 : 0;
}

extern "C" int focus_gumbel_fwd(const void* x, const float* e_soft, const float* e_hard, const void* seed, void* z,
                                int64_t* target, float* stats, int64_t R, int V, float tau, int hard, int dtype,
                                int dtype_z, void* stream) {
    if (!x || !z || !stats) return FOCUS_ERR_NULL;
    if (!focus_rows_ok(R, V, dtype) || !(tau > 0.f)) return FOCUS_ERR_SHAPE;
    if (dtype_z != dtype && !(dtype == FOCUS_F32 && dtype_z == FOCUS_BF16)) return FOCUS_ERR_DTYPE;
    const bool gen = e_soft == nullptr;
    if (gen ? seed == nullptr : (target != nullptr && e_hard == nullptr)) return FOCUS_ERR_NULL;
    if (!focus_aligned(x, 16) || !focus_aligned(z, 16) || !focus_aligned(stats, 16) || (e_soft && !focus_aligned(e_soft, 16)) ||
        (e_hard && !focus_aligned(e_hard, 16)))
        return FOCUS_ERR_ALIGN;
    GUMBEL_TZ(dtype, dtype_z, GUMBEL_NC(V, {
        if (gen)
            hipLaunchKernelGGL((gumbel_fwd_kernel<T, TZ, NC, true>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                               (const T*)x, e_soft, e_hard, (const uint32_t*)seed, (TZ*)z, target, stats, V, tau, hard);
        else
            hipLaunchKernelGGL((gumbel_fwd_kernel<T, TZ, NC, false>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                               (const T*)x, e_soft, e_hard, (const uint32_t*)seed, (TZ*)z, target, stats, V, tau, hard);
    }));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_gumbel_bwd(const void* x, const float* e_soft, const void* seed, const float* stats, const void* dz,
                                void* dx, int64_t R, int V, float tau, int dtype, int dtype_z, void* stream) {
    if (!x || !stats || !dz || !dx) return FOCUS_ERR_NULL;
    if (!focus_rows_ok(R, V, dtype) || !(tau > 0.f)) return FOCUS_ERR_SHAPE;
    if (dtype_z != dtype && !(dtype == FOCUS_F32 && dtype_z == FOCUS_BF16)) return FOCUS_ERR_DTYPE;
    const bool gen = e_soft == nullptr;
    if (gen && !seed) return FOCUS_ERR_NULL;
    if (!focus_aligned(x, 16) || !focus_aligned(dz, 16) || !focus_aligned(dx, 16) || !focus_aligned(stats, 16) ||
        (e_soft && !focus_aligned(e_soft, 16)))
        return FOCUS_ERR_ALIGN;
    GUMBEL_TZ(dtype, dtype_z, GUMBEL_NC(V, {
        if (gen)
            hipLaunchKernelGGL((gumbel_bwd_kernel<T, TZ, NC, true>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                               (const T*)x, e_soft, (const uint32_t*)seed, stats, (const TZ*)dz, (T*)dx, V, tau);
        else
            hipLaunchKernelGGL((gumbel_bwd_kernel<T, TZ, NC, false>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                               (const T*)x, e_soft, (const uint32_t*)seed, stats, (const TZ*)dz, (T*)dx, V, tau);
    }));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_xent_rows_fwd(const void* logits, const int64_t* target, float* loss_rows, float* lse, int64_t R, int V,
                                   float smoothing, int dtype, void* stream) {
    if (!logits || !target || !loss_rows || !lse) return FOCUS_ERR_NULL;
    if (!focus_rows_ok(R, V, dtype)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(logits, 16)) return FOCUS_ERR_ALIGN;
    GUMBEL_T(dtype, GUMBEL_NC(V, {
        hipLaunchKernelGGL((xent_rows_fwd_kernel<T, NC>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const T*)logits, target, loss_rows, lse, V, smoothing);
    }));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_xent_rows_bwd(const void* logits, const int64_t* target, const float* lse, const float* g,
                                   void* dlogits, int64_t R, int V, float smoothing, int dtype, void* stream) {
    if (!logits || !target || !lse || !g || !dlogits) return FOCUS_ERR_NULL;
    if (!focus_rows_ok(R, V, dtype)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(logits, 16) || !focus_aligned(dlogits, 16)) return FOCUS_ERR_ALIGN;
    GUMBEL_T(dtype, GUMBEL_NC(V, {
        hipLaunchKernelGGL((xent_rows_bwd_kernel<T, NC>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const T*)logits, target, lse, g, (T*)dlogits, V, smoothing, (float)R);
    }));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_dropout_add(const void* y, const void* res, const void* seed, int thr, void* out, int64_t n, int dtype,
                                 void* stream) {
    if (!y || !seed || !out) return FOCUS_ERR_NULL;
    if (n <= 0 || n % 8 || thr < 0 || thr >= 65536) return FOCUS_ERR_SHAPE;
    if (dtype != FOCUS_F32 && dtype != FOCUS_BF16) return FOCUS_ERR_DTYPE;
    if (!focus_aligned(y, 16) || !focus_aligned(out, 16) || (res && !focus_aligned(res, 16))) return FOCUS_ERR_ALIGN;
    const int64_t blocks = cdiv64(n / 8, 256);
    if (blocks >= (1ll << 31)) return FOCUS_ERR_SHAPE;
    GUMBEL_T(dtype, {
        hipLaunchKernelGGL((dropout_add_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)y,
                           (const T*)res, (const uint32_t*)seed, (uint32_t)thr, (T*)out, n);
    });
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
