// traj_attn.hip -- trajectory attention (attention.py:499-557).
//   * time step (fwd/bwd): one wave per (b,s); lanes cover the C channels 4 at a time, a head's d
//     channels sit on d/4 adjacent lanes, so the per-head dot products are (d/4)-lane shuffles.
//     HBM-bound: reads k2 and x~ once ([F,C] each per query), writes out [C].
//   * space step, generic path (fp32 and bf16 fallback): materialises the S x S logits in a caller
//     workspace and runs strided batched GEMMs + per-frame row softmax -- the unfused decomposition the
//     reference executes, kept as the precision path.  The fused MFMA kernel lives in traj_space_mfma.hip.
#include "focus_common.h"
#include <cstdlib>
#include "gemm_internal.h"
#include "softmax_internal.h"
#include "traj_internal.h"

namespace {

// sum over groups of `g` adjacent lanes (g a power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int g) {
    for (int o = g >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int MAXF = 16;

template <typename T, int FT>
__global__ __launch_bounds__(256) void time_fwd_kernel(const T* __restrict__ q2, const T* __restrict__ k2,
                                                       const T* __restrict__ xt, T* __restrict__ out, int64_t obs,
                                                       float* __restrict__ attn2, int64_t rows, int S, int F,
                                                       int heads, int d, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // b*S + s
    if (row >= rows) return;
    if (FT) F = FT;                       // compile-time frame count keeps lg/xv in registers
    constexpr int FA = FT ? FT : MAXF;
    const int C = heads * d, lph = d >> 2;  // lanes per head
    const int64_t b = row / S;
    const int s = (int)(row % S);
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + lane * 4;
        const bool act = c < C;
        const int h = act ? c / d : 0;
        f4 q = {0.f, 0.f, 0.f, 0.f};
        if (act) q = ld4<T>(q2 + row * C + c);
        float lg[FA];
        f4 xv[FA];
        float m = -INFINITY;
#pragma unroll
        for (int f = 0; f < (FT ? FT : F); ++f) {
            f4 kv = {0.f, 0.f, 0.f, 0.f};
            xv[f] = kv;
            if (act) {
                kv = ld4<T>(k2 + (row * F + f) * C + c);
                xv[f] = ld4<T>(xt + (row * F + f) * C + c);
            }
            lg[f] = scale * group_sum(q.x * kv.x + q.y * kv.y + q.z * kv.z + q.w * kv.w, lph);
            m = fmaxf(m, lg[f]);
        }
        float den = 0.f;
#pragma unroll
        for (int f = 0; f < (FT ? FT : F); ++f) { lg[f] = __expf(lg[f] - m); den += lg[f]; }
        const float inv = 1.f / den;
        f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < (FT ? FT : F); ++f) {
            const float a = lg[f] * inv;
            o.x += a * xv[f].x; o.y += a * xv[f].y; o.z += a * xv[f].z; o.w += a * xv[f].w;
            if (act && (lane % lph) == 0) attn2[((b * heads + h) * S + s) * F + f] = a;
        }
        if (act) st4<T>(out + b * obs + (int64_t)s * C + c, o);
    }
}

template <typename T, int FT>
__global__ __launch_bounds__(256) void time_bwd_kernel(const T* __restrict__ q2, const T* __restrict__ k2,
                                                       const T* __restrict__ xt, const float* __restrict__ attn2,
                                                       const T* __restrict__ dout, int64_t dobs, T* __restrict__ dq2,
                                                       T* __restrict__ dk2, T* __restrict__ dxt, int dxt_accum,
                                                       int64_t rows, int S, int F, int heads, int d, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (FT) F = FT;
    constexpr int FA = FT ? FT : MAXF;
    const int C = heads * d, lph = d >> 2;
    const int64_t b = row / S;
    const int s = (int)(row % S);
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + lane * 4;
        const bool act = c < C;
        const int h = act ? c / d : 0;
        f4 g = {0.f, 0.f, 0.f, 0.f}, q = g;
        if (act) { g = ld4<T>(dout + b * dobs + (int64_t)s * C + c); q = ld4<T>(q2 + row * C + c); }
        float a[FA], da[FA];
        float dot = 0.f;
#pragma unroll
        for (int f = 0; f < (FT ? FT : F); ++f) {
            a[f] = act ? attn2[((b * heads + h) * S + s) * F + f] : 0.f;
            f4 xv = {0.f, 0.f, 0.f, 0.f};
            if (act) {
                T* px = dxt + (row * F + f) * C + c;
                xv = ld4<T>(xt + (row * F + f) * C + c);
                f4 dx = {a[f] * g.x, a[f] * g.y, a[f] * g.z, a[f] * g.w};
                if (dxt_accum) { const f4 old = ld4<T>(px); dx.x += old.x; dx.y += old.y; dx.z += old.z; dx.w += old.w; }
                st4<T>(px, dx);
            }
            da[f] = group_sum(g.x * xv.x + g.y * xv.y + g.z * xv.z + g.w * xv.w, lph);
            dot += a[f] * da[f];
        }
        f4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < (FT ? FT : F); ++f) {
            const float dl = scale * a[f] * (da[f] - dot);
            if (act) {
                const f4 kv = ld4<T>(k2 + (row * F + f) * C + c);
                dq.x += dl * kv.x; dq.y += dl * kv.y; dq.z += dl * kv.z; dq.w += dl * kv.w;
                st4<T>(dk2 + (row * F + f) * C + c, (f4){dl * q.x, dl * q.y, dl * q.z, dl * q.w});
            }
        }
        if (act) st4<T>(dq2 + row * C + c, dq);
    }
}

// ---- bf16, head dim 64, F <= 16: 8 channels (16 bytes) per thread, a head = 8 adjacent lanes, thread = (row, 8-channel
// group) in memory order (C/8 threads per row; groups of 8 lanes never straddle a wave since C/8 % 8 == 0).
// Every load and store is 16 bytes and all 2F (fwd) / F (bwd) loads of a thread are in flight at once.
union Pk8 { uint4 u; bf16_t e[8]; };
__device__ __forceinline__ void unpack8(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    return o;
}
__device__ __forceinline__ float sum8lanes(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

template <int FT>
__global__ __launch_bounds__(256) void time_fwd_vec_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ k2,
                                                           const bf16_t* __restrict__ xt, bf16_t* __restrict__ out, int64_t obs,
                                                           float* __restrict__ attn2, int64_t ngroups, int S, int heads,
                                                           float scale) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool act = gi < ngroups;
    const int64_t g = act ? gi : ngroups - 1;
    const int gpr = heads * 8;                                   // 8-channel groups per row
    const int64_t row = g / gpr;
    const int cg = (int)(g - row * gpr);
    const int C = gpr * 8;
    uint4 kr[FT], xr[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        kr[f] = *reinterpret_cast<const uint4*>(k2 + ((row * FT + f) * C) + cg * 8);
        xr[f] = *reinterpret_cast<const uint4*>(xt + ((row * FT + f) * C) + cg * 8);
    }
    float q[8];
    unpack8(*reinterpret_cast<const uint4*>(q2 + row * C + cg * 8), q);
    float lg[FT], m = -INFINITY;
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        float kv[8];
        unpack8(kr[f], kv);
        float p = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) p = fmaf(q[e], kv[e], p);
        lg[f] = scale * sum8lanes(p);
        m = fmaxf(m, lg[f]);
    }
    float den = 0.f;
#pragma unroll
    for (int f = 0; f < FT; ++f) { lg[f] = __expf(lg[f] - m); den += lg[f]; }
    const float inv = 1.f / den;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t b = row / S;
    const int s = (int)(row - b * S);
    float* arow = attn2 + (((b * heads + (cg >> 3)) * S + s) * FT);
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const float a = lg[f] * inv;
        float xv[8];
        unpack8(xr[f], xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(a, xv[e], o[e]);
        if (act && (cg & 7) == 0) arow[f] = a;
    }
    if (act) *reinterpret_cast<uint4*>(out + b * obs + (int64_t)s * C + cg * 8) = pack8(o);
}

template <int FT>
__global__ __launch_bounds__(256) void time_bwd_vec_kernel(const bf16_t* __restrict__ q2, const bf16_t* __restrict__ k2,
                                                           const bf16_t* __restrict__ xt, const float* __restrict__ attn2,
                                                           const bf16_t* __restrict__ dout, int64_t dobs, bf16_t* __restrict__ dq2,
                                                           bf16_t* __restrict__ dk2, bf16_t* __restrict__ dxt,
                                                           int64_t ngroups, int S, int heads, float scale) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool act = gi < ngroups;
    const int64_t g = act ? gi : ngroups - 1;
    const int gpr = heads * 8;
    const int64_t row = g / gpr;
    const int cg = (int)(g - row * gpr);
    const int C = gpr * 8;
    const int64_t b = row / S;
    const int s = (int)(row - b * S);
    uint4 xr[FT], kr[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        xr[f] = *reinterpret_cast<const uint4*>(xt + ((row * FT + f) * C) + cg * 8);
        kr[f] = *reinterpret_cast<const uint4*>(k2 + ((row * FT + f) * C) + cg * 8);
    }
    float gv[8], q[8];
    unpack8(*reinterpret_cast<const uint4*>(dout + b * dobs + (int64_t)s * C + cg * 8), gv);
    unpack8(*reinterpret_cast<const uint4*>(q2 + row * C + cg * 8), q);
    const float* arow = attn2 + (((b * heads + (cg >> 3)) * S + s) * FT);
    float a[FT], da[FT], dot = 0.f;
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        a[f] = arow[f];
        float xv[8], dx[8];
        unpack8(xr[f], xv);
        float p = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { p = fmaf(gv[e], xv[e], p); dx[e] = a[f] * gv[e]; }
        if (act) *reinterpret_cast<uint4*>(dxt + ((row * FT + f) * C) + cg * 8) = pack8(dx);
        da[f] = sum8lanes(p);
        dot = fmaf(a[f], da[f], dot);
    }
    float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const float dl = scale * a[f] * (da[f] - dot);
        float kv[8], dk[8];
        unpack8(kr[f], kv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { dq[e] = fmaf(dl, kv[e], dq[e]); dk[e] = dl * q[e]; }
        if (act) *reinterpret_cast<uint4*>(dk2 + ((row * FT + f) * C) + cg * 8) = pack8(dk);
    }
    if (act) *reinterpret_cast<uint4*>(dq2 + row * C + cg * 8) = pack8(dq);
}

static bool time_vec_ok(const void* a, const void* b, const void* c, int F, int d, int dtype) {
    static const bool enabled = !(getenv("FOCUS_TIME_VEC") && atoi(getenv("FOCUS_TIME_VEC")) == 0);
    return enabled && dtype == FOCUS_BF16 && d == 64 && (F == 8 || F == 4 || F == 16) && focus_aligned(a, 16) &&
           focus_aligned(b, 16) && focus_aligned(c, 16);
}

// ------------------------------------------------------------------------------------------------
// generic space path
// ------------------------------------------------------------------------------------------------
struct SpaceDims {
    int B, F, P, h, d, S, N, C;
    int64_t tok;  // 3C
};

inline focus_gemm_desc base_desc(int dtype) {
    focus_gemm_desc g = {};
    g.alpha = 1.f; g.dtype_ab = dtype; g.dtype_c = dtype; g.batch0 = 1; g.batch1 = 1;
    return g;
}
inline const char* cptr(const void* p, int64_t elems, int dtype) { return (const char*)p + elems * (int64_t)focus_esize(dtype); }
inline char* mptr(void* p, int64_t elems, int dtype) { return (char*)p + elems * (int64_t)focus_esize(dtype); }

// logits[b,h,s,key] = q_[s,:].k_[key,:]   (patch tokens only)
int space_logits(const SpaceDims& D, const void* qkv, void* L, int dtype, hipStream_t s) {
    focus_gemm_desc g = base_desc(dtype);
    g.M = D.S; g.N = D.S; g.K = D.d; g.batch0 = D.B; g.batch1 = D.h;
    g.A = cptr(qkv, D.tok, dtype); g.rsA = D.tok; g.csA = 1; g.bsA0 = (int64_t)D.N * D.tok; g.bsA1 = D.d;
    g.B = cptr(qkv, D.tok + D.C, dtype); g.rsB = 1; g.csB = D.tok; g.bsB0 = (int64_t)D.N * D.tok; g.bsB1 = D.d;
    g.C = L; g.rsC = D.S; g.csC = 1; g.bsC0 = (int64_t)D.h * D.S * D.S; g.bsC1 = (int64_t)D.S * D.S;
    return focus_gemm(&g, s);
}

}  // namespace

extern "C" size_t focus_traj_space_workspace_bytes(int B, int F, int P, int heads, int d, int dtype, int backward) {
    const size_t S = (size_t)F * P, N = S + 1, C = (size_t)heads * d, es = focus_esize(dtype);
    const bool fused = focus_traj_space_mfma_ok(P, d, heads, dtype);
    size_t bytes = (size_t)B * heads * N * es;               // cls row
    bytes += focus_traj_cls_scratch_floats(B, (int)N, heads) * sizeof(float) + 256;   // cls kernels' scratch (fwd and bwd)
    if (!fused) bytes += (size_t)B * heads * S * S * es;     // logits / probabilities (unfused path)
    if (backward) {
        bytes += (size_t)B * heads * N * es + 4096;          // d(cls row) + alignment padding of the carve-up
        if (fused) {
            bytes += (size_t)2 * (B * heads * S * F * sizeof(float) + 256);   // delta * scale, lse in base-2 units
            bytes += (size_t)B * S * C * es + 256;                      // dxsum: dX rows of each query's own frame
        } else {
            bytes += (size_t)B * heads * S * S * es;         // d(prob) / d(logits)
            bytes += (size_t)B * S * F * C * es;             // dxt + diagonal term
        }
    }
    return (bytes + 255) & ~(size_t)255;
}

extern "C" int focus_traj_space_fwd(const void* qkv, void* xt, void* xdiag, void* cls_out, float* lse, float* cls_lse,
                                    void* ws, size_t ws_bytes, int B, int F, int P, int heads, int d, int dtype,
                                    void* stream) {
    if (!qkv || !xt || !xdiag || !cls_out || !lse || !cls_lse || !ws) return FOCUS_ERR_NULL;
    if (B <= 0 || F <= 0 || P <= 0 || heads <= 0 || d <= 0 || (d & 3)) return FOCUS_ERR_SHAPE;
    if (ws_bytes < focus_traj_space_workspace_bytes(B, F, P, heads, d, dtype, 0)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    SpaceDims D = {B, F, P, heads, d, F * P, F * P + 1, heads * d, 3 * (int64_t)heads * d};
    const float scale = 1.f / sqrtf((float)d);
    const bool fused = focus_traj_space_mfma_ok(P, d, heads, dtype);
    void* Lc = ws;                                           // cls row first, then the (unfused) S x S logits
    void* L = mptr(ws, (int64_t)B * heads * D.N, dtype);
    int rc;
    if (fused) {
        if ((rc = focus_traj_space_fwd_mfma(qkv, xt, xdiag, lse, B, F, P, heads, s))) return rc;
    } else {
    // patch rows: logits -> per-frame softmax (saves lse) -> per-frame A.V
    if ((rc = space_logits(D, qkv, L, dtype, s))) return rc;
    if ((rc = focus_softmax_fwd_lse(L, L, lse, (int64_t)B * heads * D.S * F, P, P, scale, dtype, s))) return rc;
    for (int f = 0; f < F; ++f) {
        focus_gemm_desc g = base_desc(dtype);
        g.M = D.S; g.N = d; g.K = P; g.batch0 = B; g.batch1 = heads;
        g.A = cptr(L, (int64_t)f * P, dtype); g.rsA = D.S; g.csA = 1;
        g.bsA0 = (int64_t)heads * D.S * D.S; g.bsA1 = (int64_t)D.S * D.S;
        g.B = cptr(qkv, (1 + (int64_t)f * P) * D.tok + 2 * D.C, dtype); g.rsB = D.tok; g.csB = 1;
        g.bsB0 = (int64_t)D.N * D.tok; g.bsB1 = d;
        g.C = mptr(xt, (int64_t)f * D.C, dtype); g.rsC = (int64_t)F * D.C; g.csC = 1;
        g.bsC0 = (int64_t)D.S * F * D.C; g.bsC1 = d;
        if ((rc = focus_gemm(&g, s))) return rc;
    }
    if ((rc = focus_diag_gather(xt, xdiag, B, D.S, F, D.C, dtype, s))) return rc;
    }
    // cls row over all N keys
    if (focus_traj_cls_ok(D.N, d)) {
        // scratch at the end of the workspace (everything before it belongs to the unfused path's logits)
        const size_t need = focus_traj_cls_scratch_floats(B, D.N, heads) * sizeof(float);
        float* scratch = reinterpret_cast<float*>(static_cast<char*>(ws) + ((ws_bytes - need) & ~(size_t)255));
        return focus_traj_cls_fwd(qkv, cls_out, cls_lse, scratch, B, D.N, heads, dtype, s);
    }
    {
        focus_gemm_desc g = base_desc(dtype);
        g.M = 1; g.N = D.N; g.K = d; g.batch0 = B; g.batch1 = heads;
        g.A = qkv; g.rsA = D.tok; g.csA = 1; g.bsA0 = (int64_t)D.N * D.tok; g.bsA1 = d;
        g.B = cptr(qkv, D.C, dtype); g.rsB = 1; g.csB = D.tok; g.bsB0 = (int64_t)D.N * D.tok; g.bsB1 = d;
        g.C = Lc; g.rsC = D.N; g.csC = 1; g.bsC0 = (int64_t)heads * D.N; g.bsC1 = D.N;
        if ((rc = focus_gemm(&g, s))) return rc;
        if ((rc = focus_softmax_fwd_lse(Lc, Lc, cls_lse, (int64_t)B * heads, D.N, D.N, scale, dtype, s))) return rc;
        focus_gemm_desc v = base_desc(dtype);
        v.M = 1; v.N = d; v.K = D.N; v.batch0 = B; v.batch1 = heads;
        v.A = Lc; v.rsA = D.N; v.csA = 1; v.bsA0 = (int64_t)heads * D.N; v.bsA1 = D.N;
        v.B = cptr(qkv, 2 * D.C, dtype); v.rsB = D.tok; v.csB = 1; v.bsB0 = (int64_t)D.N * D.tok; v.bsB1 = d;
        v.C = cls_out; v.rsC = D.C; v.csC = 1; v.bsC0 = D.C; v.bsC1 = d;
        if ((rc = focus_gemm(&v, s))) return rc;
    }
    return FOCUS_OK;
}

extern "C" int focus_traj_space_bwd(const void* qkv, const void* xt, const void* cls_out, const float* lse,
                                    const float* cls_lse, const void* dxt, const void* dxdiag, const void* dcls,
                                    void* dqkv, void* ws, size_t ws_bytes, int B, int F, int P, int heads, int d,
                                    int dtype, void* stream) {
    (void)xt; (void)cls_out;
    if (!qkv || !lse || !cls_lse || !dxt || !dxdiag || !dcls || !dqkv || !ws) return FOCUS_ERR_NULL;
    if (B <= 0 || F <= 0 || P <= 0 || heads <= 0 || d <= 0 || (d & 3)) return FOCUS_ERR_SHAPE;
    if (ws_bytes < focus_traj_space_workspace_bytes(B, F, P, heads, d, dtype, 1)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    SpaceDims D = {B, F, P, heads, d, F * P, F * P + 1, heads * d, 3 * (int64_t)heads * d};
    const float scale = 1.f / sqrtf((float)d);
    const bool fused = focus_traj_space_mfma_ok(P, d, heads, dtype);
    const int64_t nLL = fused ? 0 : (int64_t)B * heads * D.S * D.S, nLc = (int64_t)B * heads * D.N;
    const size_t es = focus_esize(dtype);
    void* Lc = ws;                        // cls probabilities
    void* dLc = mptr(Lc, (nLc + 127) & ~(int64_t)127, dtype);
    void* L = mptr(dLc, (nLc + 127) & ~(int64_t)127, dtype);   // probabilities (unfused) | delta (fused)
    void* dL = mptr(L, nLL, dtype);       // d prob -> d logits
    void* dxs = mptr(dL, nLL, dtype);     // dxt + diag(dxdiag)
    int rc;
    if (fused) {
        // patch rows first: fully writes the q/k/v parts of tokens 1..N-1; the cls step below adds onto k/v
        float* delta = reinterpret_cast<float*>(L);
        float* cls_scratch = delta + (((int64_t)B * heads * D.S * F + 63) & ~(int64_t)63);
        float* lse2 = cls_scratch + ((focus_traj_cls_scratch_floats(B, D.N, heads) + 63) & ~(size_t)63);
        void* dxsum = lse2 + (((int64_t)B * heads * D.S * F + 63) & ~(int64_t)63);
        if ((rc = focus_traj_space_bwd_mfma(qkv, xt, lse, dxt, dxdiag, delta, lse2, dxsum, dqkv, B, F, P, heads, s)))
            return rc;
    } else {
        if (hipMemcpyAsync(dxs, dxt, (size_t)B * D.S * F * D.C * es, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return FOCUS_ERR_LAUNCH;
        if ((rc = focus_diag_scatter_add(dxdiag, dxs, B, D.S, F, D.C, dtype, s))) return rc;
    }

    // ---- cls row: row 0 of every part is written plainly; rows 1.. of the k and v parts are written (unfused: the
    // patch step accumulates on top) or accumulated onto the fused kernels' output ----
    if (fused && focus_traj_cls_ok(D.N, d)) {
        float* scratch = reinterpret_cast<float*>(L) + (((int64_t)B * heads * D.S * F + 63) & ~(int64_t)63);
        return focus_traj_cls_bwd(qkv, cls_lse, dcls, dqkv, scratch, B, D.N, heads, dtype, s);
    }
    {
        focus_gemm_desc g = base_desc(dtype);   // recompute cls logits, then probabilities from lse
        g.M = 1; g.N = D.N; g.K = d; g.batch0 = B; g.batch1 = heads;
        g.A = qkv; g.rsA = D.tok; g.csA = 1; g.bsA0 = (int64_t)D.N * D.tok; g.bsA1 = d;
        g.B = cptr(qkv, D.C, dtype); g.rsB = 1; g.csB = D.tok; g.bsB0 = (int64_t)D.N * D.tok; g.bsB1 = d;
        g.C = Lc; g.rsC = D.N; g.csC = 1; g.bsC0 = (int64_t)heads * D.N; g.bsC1 = D.N;
        if ((rc = focus_gemm(&g, s))) return rc;
        if ((rc = focus_softmax_from_lse(Lc, Lc, cls_lse, (int64_t)B * heads, D.N, D.N, scale, dtype, s))) return rc;
        // dAc[n] = dcls[h,:].v[n,:]
        focus_gemm_desc a = base_desc(dtype);
        a.M = 1; a.N = D.N; a.K = d; a.batch0 = B; a.batch1 = heads;
        a.A = dcls; a.rsA = D.C; a.csA = 1; a.bsA0 = D.C; a.bsA1 = d;
        a.B = cptr(qkv, 2 * D.C, dtype); a.rsB = 1; a.csB = D.tok; a.bsB0 = (int64_t)D.N * D.tok; a.bsB1 = d;
        a.C = dLc; a.rsC = D.N; a.csC = 1; a.bsC0 = (int64_t)heads * D.N; a.bsC1 = D.N;
        if ((rc = focus_gemm(&a, s))) return rc;
        // dV[n,:] = Ac[n] * dcls[h,:]   (outer product, K=1): token 0, then tokens 1..
        for (int part = 0; part < 2; ++part) {
            focus_gemm_desc v = base_desc(dtype);
            v.M = part == 0 ? 1 : D.N - 1; v.N = d; v.K = 1; v.batch0 = B; v.batch1 = heads;
            v.A = cptr(Lc, part, dtype); v.rsA = 1; v.csA = 1; v.bsA0 = (int64_t)heads * D.N; v.bsA1 = D.N;
            v.B = dcls; v.rsB = 1; v.csB = 1; v.bsB0 = D.C; v.bsB1 = d;
            v.C = mptr(dqkv, (int64_t)part * D.tok + 2 * D.C, dtype); v.rsC = D.tok; v.csC = 1;
            v.bsC0 = (int64_t)D.N * D.tok; v.bsC1 = d;
            if (part == 1 && fused) v.residual = v.C;
            if ((rc = focus_gemm(&v, s))) return rc;
        }
        // d logits (includes the scale factor)
        if ((rc = focus_softmax_bwd(dLc, Lc, dLc, (int64_t)B * heads, D.N, D.N, scale, dtype, s))) return rc;
        // dq0[:] = sum_n dLc[n] k[n,:]
        focus_gemm_desc q = base_desc(dtype);
        q.M = 1; q.N = d; q.K = D.N; q.batch0 = B; q.batch1 = heads;
        q.A = dLc; q.rsA = D.N; q.csA = 1; q.bsA0 = (int64_t)heads * D.N; q.bsA1 = D.N;
        q.B = cptr(qkv, D.C, dtype); q.rsB = D.tok; q.csB = 1; q.bsB0 = (int64_t)D.N * D.tok; q.bsB1 = d;
        q.C = dqkv; q.rsC = D.tok; q.csC = 1; q.bsC0 = (int64_t)D.N * D.tok; q.bsC1 = d;
        if ((rc = focus_gemm(&q, s))) return rc;
        // dK[n,:] = dLc[n] * q0[:]
        for (int part = 0; part < 2; ++part) {
            focus_gemm_desc k = base_desc(dtype);
            k.M = part == 0 ? 1 : D.N - 1; k.N = d; k.K = 1; k.batch0 = B; k.batch1 = heads;
            k.A = cptr(dLc, part, dtype); k.rsA = 1; k.csA = 1; k.bsA0 = (int64_t)heads * D.N; k.bsA1 = D.N;
            k.B = qkv; k.rsB = 1; k.csB = 1; k.bsB0 = (int64_t)D.N * D.tok; k.bsB1 = d;
            k.C = mptr(dqkv, (int64_t)part * D.tok + D.C, dtype); k.rsC = D.tok; k.csC = 1;
            k.bsC0 = (int64_t)D.N * D.tok; k.bsC1 = d;
            if (part == 1 && fused) k.residual = k.C;
            if ((rc = focus_gemm(&k, s))) return rc;
        }
    }
    if (fused) return FOCUS_OK;

    // ---- patch rows ----
    if ((rc = space_logits(D, qkv, L, dtype, s))) return rc;
    if ((rc = focus_softmax_from_lse(L, L, lse, (int64_t)B * heads * D.S * F, P, P, scale, dtype, s))) return rc;
    for (int f = 0; f < F; ++f) {
        // dA[s, f, p] = dxs[s,f,h,:] . v[f*P+p, :]
        focus_gemm_desc a = base_desc(dtype);
        a.M = D.S; a.N = P; a.K = d; a.batch0 = B; a.batch1 = heads;
        a.A = cptr(dxs, (int64_t)f * D.C, dtype); a.rsA = (int64_t)F * D.C; a.csA = 1;
        a.bsA0 = (int64_t)D.S * F * D.C; a.bsA1 = d;
        a.B = cptr(qkv, (1 + (int64_t)f * P) * D.tok + 2 * D.C, dtype); a.rsB = 1; a.csB = D.tok;
        a.bsB0 = (int64_t)D.N * D.tok; a.bsB1 = d;
        a.C = mptr(dL, (int64_t)f * P, dtype); a.rsC = D.S; a.csC = 1;
        a.bsC0 = (int64_t)heads * D.S * D.S; a.bsC1 = (int64_t)D.S * D.S;
        if ((rc = focus_gemm(&a, s))) return rc;
        // dV[f*P+p, :] += sum_s A[s,f,p] dxs[s,f,h,:]
        focus_gemm_desc v = base_desc(dtype);
        v.M = P; v.N = d; v.K = D.S; v.batch0 = B; v.batch1 = heads;
        v.A = cptr(L, (int64_t)f * P, dtype); v.rsA = 1; v.csA = D.S;
        v.bsA0 = (int64_t)heads * D.S * D.S; v.bsA1 = (int64_t)D.S * D.S;
        v.B = cptr(dxs, (int64_t)f * D.C, dtype); v.rsB = (int64_t)F * D.C; v.csB = 1;
        v.bsB0 = (int64_t)D.S * F * D.C; v.bsB1 = d;
        v.C = mptr(dqkv, (1 + (int64_t)f * P) * D.tok + 2 * D.C, dtype); v.rsC = D.tok; v.csC = 1;
        v.bsC0 = (int64_t)D.N * D.tok; v.bsC1 = d;
        v.residual = v.C;   // accumulate onto the cls-row contribution
        if ((rc = focus_gemm(&v, s))) return rc;
    }
    if ((rc = focus_softmax_bwd(dL, L, dL, (int64_t)B * heads * D.S * F, P, P, scale, dtype, s))) return rc;
    {
        // dQ[s,:] = sum_key dL[s,key] k[key,:]   (rows 1.. of the q part; row 0 came from the cls step)
        focus_gemm_desc q = base_desc(dtype);
        q.M = D.S; q.N = d; q.K = D.S; q.batch0 = B; q.batch1 = heads;
        q.A = dL; q.rsA = D.S; q.csA = 1; q.bsA0 = (int64_t)heads * D.S * D.S; q.bsA1 = (int64_t)D.S * D.S;
        q.B = cptr(qkv, D.tok + D.C, dtype); q.rsB = D.tok; q.csB = 1; q.bsB0 = (int64_t)D.N * D.tok; q.bsB1 = d;
        q.C = mptr(dqkv, D.tok, dtype); q.rsC = D.tok; q.csC = 1; q.bsC0 = (int64_t)D.N * D.tok; q.bsC1 = d;
        if ((rc = focus_gemm(&q, s))) return rc;
        // dK[key,:] += sum_s dL[s,key] q[s,:]
        focus_gemm_desc k = base_desc(dtype);
        k.M = D.S; k.N = d; k.K = D.S; k.batch0 = B; k.batch1 = heads;
        k.A = dL; k.rsA = 1; k.csA = D.S; k.bsA0 = (int64_t)heads * D.S * D.S; k.bsA1 = (int64_t)D.S * D.S;
        k.B = cptr(qkv, D.tok, dtype); k.rsB = D.tok; k.csB = 1; k.bsB0 = (int64_t)D.N * D.tok; k.bsB1 = d;
        k.C = mptr(dqkv, D.tok + D.C, dtype); k.rsC = D.tok; k.csC = 1; k.bsC0 = (int64_t)D.N * D.tok; k.bsC1 = d;
        k.residual = k.C;
        if ((rc = focus_gemm(&k, s))) return rc;
    }
    return FOCUS_OK;
}

extern "C" int focus_traj_time_fwd(const void* q2, const void* k2, const void* xt, void* out, int64_t out_bstride,
                                   float* attn2, int B, int S, int F, int heads, int d, int dtype, void* stream) {
    if (!q2 || !k2 || !xt || !out || !attn2) return FOCUS_ERR_NULL;
    if (F > MAXF || F <= 0 || (d & 3) || (d >> 2) > 64 || ((d >> 2) & ((d >> 2) - 1))) return FOCUS_ERR_SHAPE;
    const int64_t rows = (int64_t)B * S;
    if (rows <= 0) return FOCUS_OK;
    const float scale = 1.f / sqrtf((float)d);
    if (out_bstride < (int64_t)S * heads * d) return FOCUS_ERR_SHAPE;
    if (time_vec_ok(q2, k2, xt, F, d, dtype) && focus_aligned(out, 16) && (out_bstride & 7) == 0) {
        const int64_t ng = rows * heads * 8;
        dim3 gv((unsigned)cdiv64(ng, 256));
#define TFV(FT) hipLaunchKernelGGL((time_fwd_vec_kernel<FT>), gv, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q2, (const bf16_t*)k2, (const bf16_t*)xt, (bf16_t*)out, out_bstride, attn2, ng, S, heads, scale)
        if (F == 8) TFV(8); else if (F == 4) TFV(4); else TFV(16);
#undef TFV
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    dim3 grid((unsigned)cdiv64(rows, 4));
#define TF(T, FT) hipLaunchKernelGGL((time_fwd_kernel<T, FT>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q2, (const T*)k2, (const T*)xt, (T*)out, out_bstride, attn2, rows, S, F, heads, d, scale)
    if (dtype == FOCUS_BF16) { if (F == 8) TF(bf16_t, 8); else TF(bf16_t, 0); }
    else { if (F == 8) TF(float, 8); else TF(float, 0); }
#undef TF
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_traj_time_bwd(const void* q2, const void* k2, const void* xt, const float* attn2,
                                   const void* dout, int64_t dout_bstride, void* dq2, void* dk2, void* dxt, int dxt_accum,
                                   int B, int S, int F, int heads, int d, int dtype, void* stream) {
    if (!q2 || !k2 || !xt || !attn2 || !dout || !dq2 || !dk2 || !dxt) return FOCUS_ERR_NULL;
    if (F > MAXF || F <= 0 || (d & 3) || (d >> 2) > 64 || ((d >> 2) & ((d >> 2) - 1))) return FOCUS_ERR_SHAPE;
    const int64_t rows = (int64_t)B * S;
    if (rows <= 0) return FOCUS_OK;
    const float scale = 1.f / sqrtf((float)d);
    if (dout_bstride < (int64_t)S * heads * d) return FOCUS_ERR_SHAPE;
    if (!dxt_accum && (dout_bstride & 7) == 0 && time_vec_ok(q2, k2, xt, F, d, dtype) && focus_aligned(dout, 16) && focus_aligned(dq2, 16) &&
        focus_aligned(dk2, 16) && focus_aligned(dxt, 16)) {
        const int64_t ng = rows * heads * 8;
        dim3 gv((unsigned)cdiv64(ng, 256));
#define TBV(FT) hipLaunchKernelGGL((time_bwd_vec_kernel<FT>), gv, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q2, (const bf16_t*)k2, (const bf16_t*)xt, attn2, (const bf16_t*)dout, dout_bstride, (bf16_t*)dq2, (bf16_t*)dk2, (bf16_t*)dxt, ng, S, heads, scale)
        if (F == 8) TBV(8); else if (F == 4) TBV(4); else TBV(16);
#undef TBV
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    dim3 grid((unsigned)cdiv64(rows, 4));
#define TB(T, FT) hipLaunchKernelGGL((time_bwd_kernel<T, FT>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q2, (const T*)k2, (const T*)xt, attn2, (const T*)dout, dout_bstride, (T*)dq2, (T*)dk2, (T*)dxt, dxt_accum, rows, S, F, heads, d, scale)
    if (dtype == FOCUS_BF16) { if (F == 8) TB(bf16_t, 8); else TB(bf16_t, 0); }
    else { if (F == 8) TB(float, 8); else TB(float, 0); }
#undef TB
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
