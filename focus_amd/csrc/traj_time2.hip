// traj_time2.hip -- temporal step of trajectory attention (attention.py:538-549) in re-associated form:
//   logits[s,f,h] = scale * u[s,h,:] . x~[s,f,:],  u[s,h,:] = Wk[h]^T q2[s,h,:]   (see include/focus_amd.h)
// One wave per (b,s): the 8 x C slab of x~ is read ONCE into registers (4 channels per lane per 256-channel chunk),
// each head's 8 frame logits are 8 wave reductions of C-long dot products, softmax over frames, and the output /
// gradients are finished from the registers.  HBM-bound: x~ (F*C) + u (h*C) read, out (C) written per query.
#include "focus_common.h"

namespace {

constexpr int FT = 8;        // frames (compile-time: keeps x~ in registers); other F use the k2 path
constexpr int NCH = 3;       // 256-channel chunks handled per lane (C <= 768)

template <typename T>
__global__ __launch_bounds__(256) void time2_fwd_kernel(const T* __restrict__ u, const T* __restrict__ xt,
                                                        T* __restrict__ out, float* __restrict__ attn2, int64_t rows,
                                                        int S, int heads, int d, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C = heads * d, lph = d >> 2;
    const int64_t b = row / S;
    const int s = (int)(row % S);
    f4 x[FT][NCH];
    bool act[NCH];
    int hd[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + lane * 4;
        act[i] = c < C;
        hd[i] = act[i] ? c / d : -1;
#pragma unroll
        for (int f = 0; f < FT; ++f)
            x[f][i] = act[i] ? ld4<T>(xt + (row * FT + f) * C + c) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    float a[NCH][FT];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int f = 0; f < FT; ++f) a[i][f] = 0.f;
    for (int h = 0; h < heads; ++h) {
        f4 uv[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            uv[i] = act[i] ? ld4<T>(u + (row * heads + h) * C + i * 256 + lane * 4) : (f4){0.f, 0.f, 0.f, 0.f};
        float lg[FT], m = -INFINITY;
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                p += uv[i].x * x[f][i].x + uv[i].y * x[f][i].y + uv[i].z * x[f][i].z + uv[i].w * x[f][i].w;
            lg[f] = scale * wave_sum(p);
            m = fmaxf(m, lg[f]);
        }
        float den = 0.f;
#pragma unroll
        for (int f = 0; f < FT; ++f) { lg[f] = __expf(lg[f] - m); den += lg[f]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bool mine = hd[i] == h;
#pragma unroll
            for (int f = 0; f < FT; ++f) a[i][f] = mine ? lg[f] * inv : a[i][f];
        }
#pragma unroll
        for (int f = 0; f < FT; ++f)
            if (lane == f) attn2[((b * heads + h) * S + s) * FT + f] = lg[f] * inv;
    }
    (void)lph;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (!act[i]) continue;
        f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            o.x += a[i][f] * x[f][i].x; o.y += a[i][f] * x[f][i].y; o.z += a[i][f] * x[f][i].z; o.w += a[i][f] * x[f][i].w;
        }
        st4<T>(out + row * C + i * 256 + lane * 4, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void time2_bwd_kernel(const T* __restrict__ u, const T* __restrict__ xt,
                                                        const float* __restrict__ attn2, const T* __restrict__ dout,
                                                        T* __restrict__ du, T* __restrict__ dxt, int dxt_accum,
                                                        int64_t rows, int S, int heads, int d, float scale) {
    __shared__ float sdl[4][16 * FT];                 // per wave: dlogit[h][f] (scale included)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    const bool valid = row < rows;                    // no early return: the wave-level LDS exchange needs every lane
    const int64_t rw = valid ? row : rows - 1;
    const int C = heads * d, lph = d >> 2;
    const int64_t b = rw / S;
    const int s = (int)(rw % S);
    f4 x[FT][NCH], g[NCH];
    bool act[NCH];
    int hd[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + lane * 4;
        act[i] = c < C;
        hd[i] = act[i] ? c / d : 0;
        g[i] = act[i] ? ld4<T>(dout + rw * C + c) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FT; ++f)
            x[f][i] = act[i] ? ld4<T>(xt + (rw * FT + f) * C + c) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    // d attn[f] for this lane's head (reduce over the head's d/4 lanes), then d logits
    float a[NCH][FT];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        float da[FT], dot = 0.f;
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            a[i][f] = act[i] ? attn2[((b * heads + hd[i]) * S + s) * FT + f] : 0.f;
            float p = g[i].x * x[f][i].x + g[i].y * x[f][i].y + g[i].z * x[f][i].z + g[i].w * x[f][i].w;
            for (int o = lph >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            da[f] = p;
            dot += a[i][f] * p;
        }
        if (act[i] && (lane % lph) == 0) {
#pragma unroll
            for (int f = 0; f < FT; ++f) sdl[w][hd[i] * FT + f] = scale * a[i][f] * (da[f] - dot);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // dxt = attn * dout + sum_h dlogit[f,h] * u[h]  ;  du[h] = sum_f dlogit[f,h] * x[f]
    f4 dx[FT][NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int f = 0; f < FT; ++f)
            dx[f][i] = {a[i][f] * g[i].x, a[i][f] * g[i].y, a[i][f] * g[i].z, a[i][f] * g[i].w};
    for (int h = 0; h < heads; ++h) {
        float dl[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) dl[f] = sdl[w][h * FT + f];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (!act[i]) continue;
            const int64_t uo = (rw * heads + h) * C + i * 256 + lane * 4;
            const f4 uv = ld4<T>(u + uo);
            f4 dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                dx[f][i].x += dl[f] * uv.x; dx[f][i].y += dl[f] * uv.y; dx[f][i].z += dl[f] * uv.z; dx[f][i].w += dl[f] * uv.w;
                dv.x += dl[f] * x[f][i].x; dv.y += dl[f] * x[f][i].y; dv.z += dl[f] * x[f][i].z; dv.w += dl[f] * x[f][i].w;
            }
            if (valid) st4<T>(du + uo, dv);
        }
    }
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (!act[i]) continue;
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            T* p = dxt + (rw * FT + f) * C + i * 256 + lane * 4;
            f4 o = dx[f][i];
            if (dxt_accum) { const f4 old = ld4<T>(p); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            st4<T>(p, o);
        }
    }
}

bool shape_ok(int F, int heads, int d) {
    const int lph = d >> 2;
    return F == FT && (d & 3) == 0 && lph >= 1 && lph <= 64 && (lph & (lph - 1)) == 0 && heads >= 1 && heads <= 16 &&
           heads * d <= NCH * 256;
}

}  // namespace

extern "C" int focus_traj_time2_fwd(const void* u, const void* xt, void* out, float* attn2, int B, int S, int F,
                                    int heads, int d, int dtype, void* stream) {
    if (!u || !xt || !out || !attn2) return FOCUS_ERR_NULL;
    if (!shape_ok(F, heads, d)) return FOCUS_ERR_SHAPE;
    const int64_t rows = (int64_t)B * S;
    if (rows <= 0) return FOCUS_OK;
    const float scale = 1.f / sqrtf((float)d);
    dim3 grid((unsigned)cdiv64(rows, 4));
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((time2_fwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u,
                           (const bf16_t*)xt, (bf16_t*)out, attn2, rows, S, heads, d, scale);
    else
        hipLaunchKernelGGL((time2_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)u,
                           (const float*)xt, (float*)out, attn2, rows, S, heads, d, scale);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_traj_time2_bwd(const void* u, const void* xt, const float* attn2, const void* dout, void* du,
                                    void* dxt, int dxt_accum, int B, int S, int F, int heads, int d, int dtype,
                                    void* stream) {
    if (!u || !xt || !attn2 || !dout || !du || !dxt) return FOCUS_ERR_NULL;
    if (!shape_ok(F, heads, d)) return FOCUS_ERR_SHAPE;
    const int64_t rows = (int64_t)B * S;
    if (rows <= 0) return FOCUS_OK;
    const float scale = 1.f / sqrtf((float)d);
    dim3 grid((unsigned)cdiv64(rows, 4));
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((time2_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u,
                           (const bf16_t*)xt, attn2, (const bf16_t*)dout, (bf16_t*)du, (bf16_t*)dxt, dxt_accum, rows, S,
                           heads, d, scale);
    else
        hipLaunchKernelGGL((time2_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)u,
                           (const float*)xt, attn2, (const float*)dout, (float*)du, (float*)dxt, dxt_accum, rows, S, heads,
                           d, scale);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
