// orvit_tokens.hip -- token plumbing of the ORViT block (ORViT/orvit.py:145-147, 152-157, 165-169) as single passes.
//
// The reference builds the block's token sequence with two torch.cat, takes the attention output apart with a slice + a
// reshape copy, re-attaches the cls row with a third cat and adds the result to the residual stream; autograd then runs
// slice_backward / cat_backward / add chains over the same 58 MB tensors (measured: ~2 ms of ATen kernels per bench step
// for the three ORViT blocks).  Here each direction of each step is one HBM pass:
//   assemble      all[b, 0] = x[b, 0];  all[b, 1 + t (HW+O) + p] = p < HW ? x[b, 1 + t HW + p] : obj[b, t, p - HW]
//   assemble_bwd  the same map run backwards: d(all) rows -> dx rows (all of them) and dobj rows
//   merge         out = x + s_b (gather(y) + [0; mm])          y: attention output over `all`, mm: motion-stream MLP
//   merge_bwd     dy = s_b scatter(dout) (object rows zero), dmm = s_b dout[:, 1:]     (dx is dout itself)
// Rows are moved in 16-byte pieces, one piece per thread; a workgroup covers 256 consecutive pieces of the LARGER
// tensor of the pass, so every access is a whole-row-segment stream (no LDS, nothing to tile: HBM copy roofline).
#include "focus_common.h"

namespace {

struct RowMap { int b, t, p; bool cls; };
// row r of a [B, 1 + T*(HW+O), C] token buffer
__device__ __forceinline__ RowMap map_all_row(int64_t r, int T, int HW, int O) {
    const int per = 1 + T * (HW + O);
    RowMap m;
    m.b = (int)(r / per);
    const int lr = (int)(r - (int64_t)m.b * per);
    m.cls = lr == 0;
    m.t = m.cls ? 0 : (lr - 1) / (HW + O);
    m.p = m.cls ? 0 : (lr - 1) - m.t * (HW + O);
    return m;
}

// pure copies (dtype agnostic): FWD  all <- (x, obj);  !FWD  (dx, dobj) <- dall
template <bool FWD>
__global__ __launch_bounds__(256) void orvit_assemble_kernel(const uint4* __restrict__ a, const uint4* __restrict__ o_in,
                                                             uint4* __restrict__ out_a, uint4* __restrict__ out_o, int64_t npieces,
                                                             int vpr, int T, int HW, int O) {
    // FWD: a = x, o_in = obj, out_a = all.   !FWD: a = dall, out_a = dx, out_o = dobj.
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;    // piece index in the `all`-shaped tensor
    if (i >= npieces) return;
    const int64_t r = i / vpr;
    const int v = (int)(i - r * vpr);
    const RowMap m = map_all_row(r, T, HW, O);
    const int64_t xrow = (int64_t)m.b * (1 + T * HW) + (m.cls ? 0 : 1 + m.t * HW + m.p);
    const int64_t orow = ((int64_t)m.b * T + m.t) * O + (m.p - HW);
    const bool is_obj = !m.cls && m.p >= HW;
    if (FWD) out_a[i] = is_obj ? o_in[orow * vpr + v] : a[xrow * vpr + v];
    else if (is_obj) out_o[orow * vpr + v] = a[i];
    else out_a[xrow * vpr + v] = a[i];
}

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };
template <typename T> __device__ __forceinline__ void unpack(const uint4& r, float* v);
template <> __device__ __forceinline__ void unpack<float>(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack<bf16_t>(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack(const float* v);
template <> __device__ __forceinline__ uint4 pack<float>(const float* v) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 pack<bf16_t>(const float* v) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    return o;
}

// out [B, 1+T*HW, C]: one piece of `out` per thread
template <typename T>
__global__ __launch_bounds__(256) void orvit_merge_kernel(const uint4* __restrict__ x, const uint4* __restrict__ y,
                                                          const uint4* __restrict__ mm, const float* __restrict__ scale,
                                                          uint4* __restrict__ out, int64_t npieces, int vpr, int Tn, int HW,
                                                          int O) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npieces) return;
    const int64_t r = i / vpr;
    const int v = (int)(i - r * vpr);
    const int per = 1 + Tn * HW;
    const int b = (int)(r / per), lr = (int)(r - (int64_t)b * per);
    const int t = lr ? (lr - 1) / HW : 0, p = lr ? (lr - 1) - t * HW : 0;
    const int64_t yrow = (int64_t)b * (1 + Tn * (HW + O)) + (lr ? 1 + t * (HW + O) + p : 0);
    constexpr int N = Vec<T>::N;
    float xv[N], yv[N];
    unpack<T>(x[i], xv);
    unpack<T>(y[yrow * vpr + v], yv);
    if (mm && lr) {
        float mv[N];
        unpack<T>(mm[((int64_t)b * Tn * HW + (lr - 1)) * vpr + v], mv);
#pragma unroll
        for (int e = 0; e < N; ++e) yv[e] += mv[e];
    }
    const float s = scale ? scale[b] : 1.f;
#pragma unroll
    for (int e = 0; e < N; ++e) xv[e] = fmaf(s, yv[e], xv[e]);
    out[i] = pack<T>(xv);
}

// dy [B, 1+T*(HW+O), C]: one piece of dy per thread; the thread of a patch row also writes that row of dmm
template <typename T>
__global__ __launch_bounds__(256) void orvit_merge_bwd_kernel(const uint4* __restrict__ dout, const float* __restrict__ scale,
                                                              uint4* __restrict__ dy, uint4* __restrict__ dmm, int64_t npieces,
                                                              int vpr, int Tn, int HW, int O) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npieces) return;
    const int64_t r = i / vpr;
    const int v = (int)(i - r * vpr);
    const RowMap m = map_all_row(r, Tn, HW, O);
    if (!m.cls && m.p >= HW) { dy[i] = make_uint4(0, 0, 0, 0); return; }
    const int lr = m.cls ? 0 : 1 + m.t * HW + m.p;
    uint4 g = dout[((int64_t)m.b * (1 + Tn * HW) + lr) * vpr + v];
    if (scale) {
        constexpr int N = Vec<T>::N;
        float gv[N];
        unpack<T>(g, gv);
        const float s = scale[m.b];
#pragma unroll
        for (int e = 0; e < N; ++e) gv[e] *= s;
        g = pack<T>(gv);
    }
    dy[i] = g;
    if (dmm && lr) dmm[((int64_t)m.b * Tn * HW + (lr - 1)) * vpr + v] = g;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int esize(int dtype) { return dtype == FOCUS_BF16 ? 2 : 4; }
inline bool shape_ok(int B, int T, int HW, int O, int C, int dtype) {
    return B > 0 && T > 0 && HW > 0 && O >= 0 && C > 0 && (dtype == FOCUS_BF16 || dtype == FOCUS_F32) && (C * esize(dtype)) % 16 == 0 &&
           (int64_t)B * (1 + (int64_t)T * (HW + O)) * C < (1LL << 40);
}

}  // namespace

extern "C" int focus_orvit_assemble(const void* x, const void* obj, void* all, int B, int T, int HW, int O, int C, int dtype,
                                    void* stream) {
    if (!x || !all || (O > 0 && !obj)) return FOCUS_ERR_NULL;
    if (!shape_ok(B, T, HW, O, C, dtype)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(x, 16) || !focus_aligned(all, 16) || (obj && !focus_aligned(obj, 16))) return FOCUS_ERR_ALIGN;
    const int vpr = C * esize(dtype) / 16;
    const int64_t n = (int64_t)B * (1 + (int64_t)T * (HW + O)) * vpr;
    hipLaunchKernelGGL((orvit_assemble_kernel<true>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, (const uint4*)obj, (uint4*)all, (uint4*)nullptr, n, vpr, T, HW, O);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_orvit_assemble_bwd(const void* dall, void* dx, void* dobj, int B, int T, int HW, int O, int C, int dtype,
                                        void* stream) {
    if (!dall || !dx || (O > 0 && !dobj)) return FOCUS_ERR_NULL;
    if (!shape_ok(B, T, HW, O, C, dtype)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(dall, 16) || !focus_aligned(dx, 16) || (dobj && !focus_aligned(dobj, 16))) return FOCUS_ERR_ALIGN;
    const int vpr = C * esize(dtype) / 16;
    const int64_t n = (int64_t)B * (1 + (int64_t)T * (HW + O)) * vpr;
    hipLaunchKernelGGL((orvit_assemble_kernel<false>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)dall, (const uint4*)nullptr, (uint4*)dx, (uint4*)dobj, n, vpr, T, HW, O);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_orvit_merge(const void* x, const void* y, const void* mm, const float* scale, void* out, int B, int T,
                                 int HW, int O, int C, int dtype, void* stream) {
    if (!x || !y || !out) return FOCUS_ERR_NULL;
    if (!shape_ok(B, T, HW, O, C, dtype)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(x, 16) || !focus_aligned(y, 16) || !focus_aligned(out, 16) || (mm && !focus_aligned(mm, 16)))
        return FOCUS_ERR_ALIGN;
    const int vpr = C * esize(dtype) / 16;
    const int64_t n = (int64_t)B * (1 + (int64_t)T * HW) * vpr;
    const dim3 grid((unsigned)cdiv(n, 256));
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((orvit_merge_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)y,
                           (const uint4*)mm, scale, (uint4*)out, n, vpr, T, HW, O);
    else
        hipLaunchKernelGGL((orvit_merge_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)y,
                           (const uint4*)mm, scale, (uint4*)out, n, vpr, T, HW, O);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_orvit_merge_bwd(const void* dout, const float* scale, void* dy, void* dmm, int B, int T, int HW, int O,
                                     int C, int dtype, void* stream) {
    if (!dout || !dy) return FOCUS_ERR_NULL;
    if (!shape_ok(B, T, HW, O, C, dtype)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(dout, 16) || !focus_aligned(dy, 16) || (dmm && !focus_aligned(dmm, 16))) return FOCUS_ERR_ALIGN;
    const int vpr = C * esize(dtype) / 16;
    const int64_t n = (int64_t)B * (1 + (int64_t)T * (HW + O)) * vpr;
    const dim3 grid((unsigned)cdiv(n, 256));
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((orvit_merge_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, scale,
                           (uint4*)dy, (uint4*)dmm, n, vpr, T, HW, O);
    else
        hipLaunchKernelGGL((orvit_merge_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, scale,
                           (uint4*)dy, (uint4*)dmm, n, vpr, T, HW, O);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
