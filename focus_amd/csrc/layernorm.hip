// layernorm.hip -- LayerNorm forward/backward, one wave (64 lanes) per row, 4-wide vector access.
// HBM-bound: fwd reads x once and writes y once (2*D*esize B/row); bwd reads dy,x once, writes dx once.
#include "focus_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int MAXV = 16;  // up to 16 x (64 lanes x 4) = 4096 columns

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int D, float eps, int rpb, int64_t xbs) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (int64_t)(row / rpb) * xbs + (int64_t)(row % rpb) * D;   // row blocks of rpb rows, xbs elements apart
    f4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) { v[i] = ld4<T>(xr + c); s += v[i].x + v[i].y + v[i].z + v[i].w; }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
            q += a * a + b * b + cc * cc + dd * dd;
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    T* yr = y + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const f4 g = ld4<float>(gamma + c), b = ld4<float>(beta + c);
            f4 o = {(v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y,
                    (v[i].z - mu) * rs * g.z + b.z, (v[i].w - mu) * rs * g.w + b.w};
            st4<T>(yr + c, o);
        }
    }
}

// Each block walks rows blockIdx.x*4+w, += gridDim.x*4; per-lane partial dgamma/dbeta kept in registers,
// combined across the block's 4 waves through LDS, written to partial[0|1][blk][D].
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ partial, int rows, int D,
                                                     int rpb, int64_t xbs) {
    __shared__ float red[4][64 * 4 + 4];
    // x and dx live in row blocks of rpb rows, xbs elements apart (dense: rpb = rows); dy and dres are dense
    auto xoff = [&](int row) __attribute__((always_inline)) { return (int64_t)(row / rpb) * xbs + (int64_t)(row % rpb) * D; };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4 dg[NV], db[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = {0.f, 0.f, 0.f, 0.f};
        db[i] = {0.f, 0.f, 0.f, 0.f};
        const int c = (i * 64 + lane) * 4;
        g[i] = c < D ? ld4<float>(gamma + c) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    // two rows per iteration: both rows' loads are issued before either is consumed (the kernel is latency-bound at
    // a few waves per SIMD; one row at a time measured 44 us for 12552x768 = 1.3 TB/s)
    const int stride = gridDim.x * 4;
    for (int row0 = blockIdx.x * 4 + w; row0 < rows; row0 += 2 * stride) {
        const int row1 = row0 + stride;
        const bool two = row1 < rows;
        const int rowB = two ? row1 : row0;
        const float mu0 = mean[row0], rs0 = rstd[row0], mu1 = mean[rowB], rs1 = rstd[rowB];
        f4 xa[NV], da[NV], xb[NV], dbv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                xa[i] = ld4<T>(x + xoff(row0) + c); da[i] = ld4<T>(dy + (int64_t)row0 * D + c);
                xb[i] = ld4<T>(x + xoff(rowB) + c); dbv[i] = ld4<T>(dy + (int64_t)rowB * D + c);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            if (rr == 1 && !two) break;
            const float mu = rr ? mu1 : mu0, rs = rr ? rs1 : rs0;
            f4 xh[NV], gd[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    const f4 xv = rr ? xb[i] : xa[i], dv = rr ? dbv[i] : da[i];
                    xh[i] = {(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
                    gd[i] = {dv.x * g[i].x, dv.y * g[i].y, dv.z * g[i].z, dv.w * g[i].w};
                    s1 += gd[i].x + gd[i].y + gd[i].z + gd[i].w;
                    s2 += gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w;
                    dg[i].x += dv.x * xh[i].x; dg[i].y += dv.y * xh[i].y; dg[i].z += dv.z * xh[i].z; dg[i].w += dv.w * xh[i].w;
                    db[i].x += dv.x; db[i].y += dv.y; db[i].z += dv.z; db[i].w += dv.w;
                }
            }
            const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
            T* dxr = dx + xoff(rr ? row1 : row0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f4 o = {rs * (gd[i].x - m1 - xh[i].x * m2), rs * (gd[i].y - m1 - xh[i].y * m2),
                            rs * (gd[i].z - m1 - xh[i].z * m2), rs * (gd[i].w - m1 - xh[i].w * m2)};
                    if (dres) {      // gradient arriving on the residual path around this LayerNorm
                        const f4 e = ld4<T>(dres + (int64_t)(rr ? row1 : row0) * D + c);
                        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                    }
                    st4<T>(dxr + c, o);
                }
            }
        }
    }
    float* pg = partial + (int64_t)blockIdx.x * D;
    float* pb = partial + (int64_t)(gridDim.x + blockIdx.x) * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        for (int pass = 0; pass < 2; ++pass) {
            const f4 val = pass == 0 ? dg[i] : db[i];
            __syncthreads();
            red[w][lane * 4 + 0] = val.x; red[w][lane * 4 + 1] = val.y;
            red[w][lane * 4 + 2] = val.z; red[w][lane * 4 + 3] = val.w;
            __syncthreads();
            if (w == 0) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f4 o;
                    o.x = red[0][lane * 4 + 0] + red[1][lane * 4 + 0] + red[2][lane * 4 + 0] + red[3][lane * 4 + 0];
                    o.y = red[0][lane * 4 + 1] + red[1][lane * 4 + 1] + red[2][lane * 4 + 1] + red[3][lane * 4 + 1];
                    o.z = red[0][lane * 4 + 2] + red[1][lane * 4 + 2] + red[2][lane * 4 + 2] + red[3][lane * 4 + 2];
                    o.w = red[0][lane * 4 + 3] + red[1][lane * 4 + 3] + red[2][lane * 4 + 3] + red[3][lane * 4 + 3];
                    st4<float>((pass == 0 ? pg : pb) + c, o);
                }
            }
        }
    }
}

// grid (ceil(D/64), 2), 1024 threads = 64 columns x 16 row-lanes; blockIdx.y selects dgamma / dbeta.  A wave reads
// 256 contiguous bytes of a partial row (16 columns per block meant 64-byte pieces: 9.7 us for the 3 MB of D = 768).
__global__ __launch_bounds__(1024) void ln_bwd_finish(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, int nblk, int D) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float* p = partial + (int64_t)blockIdx.y * nblk * D;
    float a = 0.f;
    if (c < D) {
#pragma unroll 8                                                      // (eight partial rows in flight per thread)
        for (int k = rl; k < nblk; k += 16) a += p[(int64_t)k * D + c];
    }
    red[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        (blockIdx.y == 0 ? dgamma : dbeta)[c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// bf16, D % 8 == 0, D <= 1024: 16 bytes per lane and a SUB-WAVE per row.  The kernels above give a row a whole wave and
// 8 bytes per lane: at D = 192 (the STEVE tokens, 131072 rows per frame) 48 lanes issue 8-byte loads -- 46 / 80 us per
// call forward / backward against 12.5 / 19 us of HBM time.  Here a row takes LPR = 8, 16, 32 or 64 lanes (the smallest
// with LPR * 8 * NV >= D), a wave holds 64 / LPR rows per pass and keeps UN passes in flight; the row reductions are
// log2(LPR) xor-shuffle steps.  D = 768 (ORViT tokens) runs as LPR = 64, NV = 2.
// ------------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(f[0]) | ((uint32_t)f32_to_bf16(f[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(f[2]) | ((uint32_t)f32_to_bf16(f[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(f[4]) | ((uint32_t)f32_to_bf16(f[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(f[6]) | ((uint32_t)f32_to_bf16(f[7]) << 16);
    return o;
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <int LPR, int NV>
__global__ __launch_bounds__(256) void ln_fwd_v16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                         float eps, int rpb, int64_t xbs) {
    constexpr int RPW = 64 / LPR, UN = NV == 1 ? 4 : 2;
    const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR;
    float g[NV][8], bt[NV][8];
    bool act[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * LPR + sl) * 8;
        act[i] = c < D;
        if (act[i]) { load8f(gamma + c, g[i]); load8f(beta + c, bt[i]); }
    }
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const float invD = 1.f / (float)D;
    for (int r0 = wave * (RPW * UN); r0 < rows; r0 += nwaves * (RPW * UN)) {
        uint4 v[UN][NV];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = min(r0 + u * RPW + sub, rows - 1);
            const bf16_t* xr = x + (int64_t)(row / rpb) * xbs + (int64_t)(row % rpb) * D;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                v[u][i] = act[i] ? *reinterpret_cast<const uint4*>(xr + (i * LPR + sl) * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = r0 + u * RPW + sub;
            float f[NV][8], s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                unpack8(v[u][i], f[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[i][e];
            }
            const float mu = sub_sum<LPR>(s) * invD;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (act[i]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { f[i][e] -= mu; q += f[i][e] * f[i][e]; }
                }
            const float rs = rsqrtf(sub_sum<LPR>(q) * invD + eps);
            if (row < rows) {
                if (sl == 0) { mean[row] = mu; rstd[row] = rs; }
                bf16_t* yr = y + (int64_t)row * D;
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (act[i]) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = f[i][e] * rs * g[i][e] + bt[i][e];
                        *reinterpret_cast<uint4*>(yr + (i * LPR + sl) * 8) = pack8(o);
                    }
            }
        }
    }
}

// Backward: per-lane partial dgamma/dbeta over every row the lane's sub-wave visits, combined across the workgroup's
// 4 * (64 / LPR) sub-waves through LDS into partial[0|1][blk][D] (ln_bwd_finish sums the blocks).
template <int LPR, int NV, int NW>     // NW waves per workgroup
__global__ __launch_bounds__(64 * NW) void ln_bwd_v16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                         bf16_t* __restrict__ dx, float* __restrict__ partial, int rows, int D,
                                                         int rpb, int64_t xbs) {
    constexpr int RPW = 64 / LPR, UN = NV == 1 ? 4 : 2, NSUB = NW * RPW;
    extern __shared__ float red[];                                // [2][NSUB][D]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane / LPR, sl = lane % LPR;
    float g[NV][8], dg[NV][8], db[NV][8];
    bool act[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * LPR + sl) * 8;
        act[i] = c < D;
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[i][e] = 0.f; dg[i][e] = 0.f; db[i][e] = 0.f; }
        if (act[i]) load8f(gamma + c, g[i]);
    }
    const int wave = blockIdx.x * NW + w, nwaves = gridDim.x * NW;
    const float invD = 1.f / (float)D;
    for (int r0 = wave * (RPW * UN); r0 < rows; r0 += nwaves * (RPW * UN)) {
        uint4 xv[UN][NV], dv[UN][NV];
        float mu[UN], rs[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = min(r0 + u * RPW + sub, rows - 1);
            const bf16_t* xr = x + (int64_t)(row / rpb) * xbs + (int64_t)(row % rpb) * D;
            const bf16_t* dr = dy + (int64_t)row * D;
            mu[u] = mean[row]; rs[u] = rstd[row];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * LPR + sl) * 8;
                xv[u][i] = act[i] ? *reinterpret_cast<const uint4*>(xr + c) : make_uint4(0, 0, 0, 0);
                dv[u][i] = act[i] ? *reinterpret_cast<const uint4*>(dr + c) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = r0 + u * RPW + sub;
            const bool valid = row < rows;
            float xh[NV][8], gd[NV][8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float d[8];
                unpack8(xv[u][i], xh[i]);
                unpack8(dv[u][i], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[i][e] = act[i] ? (xh[i][e] - mu[u]) * rs[u] : 0.f;
                    gd[i][e] = d[e] * g[i][e];
                    s1 += gd[i][e];
                    s2 += gd[i][e] * xh[i][e];
                    if (valid) { dg[i][e] += d[e] * xh[i][e]; db[i][e] += d[e]; }
                }
            }
            const float m1 = sub_sum<LPR>(s1) * invD, m2 = sub_sum<LPR>(s2) * invD;
            if (valid) {
                bf16_t* dxr = dx + (int64_t)(row / rpb) * xbs + (int64_t)(row % rpb) * D;
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (act[i]) {
                        const int c = (i * LPR + sl) * 8;
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rs[u] * (gd[i][e] - m1 - xh[i][e] * m2);
                        if (dres) {          // gradient arriving on the residual path around this LayerNorm
                            float r[8];
                            unpack8(*reinterpret_cast<const uint4*>(dres + (int64_t)row * D + c), r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += r[e];
                        }
                        *reinterpret_cast<uint4*>(dxr + c) = pack8(o);
                    }
            }
        }
    }
    float* rg = red + (size_t)(w * RPW + sub) * D;
    float* rb = red + (size_t)(NSUB + w * RPW + sub) * D;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (act[i]) {
            const int c = (i * LPR + sl) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { rg[c + e] = dg[i][e]; rb[c + e] = db[i][e]; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 64 * NW) {
        const int which = c >= D, col = which ? c - D : c;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NSUB; ++k) t += red[(size_t)(which * NSUB + k) * D + col];
        partial[(int64_t)(which * gridDim.x + blockIdx.x) * D + col] = t;
    }
}

// (rows < 4096 -- the per-slot LayerNorms of STEVE, 352 rows -- stay with the wave-per-row kernels: a launch that small
// is a fixed few microseconds either way and the sub-wave kernels' wider LDS combine measured 1 us slower there)
inline bool ln_v16_ok(const void* a, const void* b, const void* c, const void* d, int rows, int D, int64_t xbs, int dtype) {
    static const bool enabled = !(getenv("FOCUS_LN_V16") && atoi(getenv("FOCUS_LN_V16")) == 0);
    return enabled && rows >= 4096 && dtype == FOCUS_BF16 && (D & 7) == 0 && D <= 1024 && (xbs & 7) == 0 && focus_aligned(a, 16) &&
           focus_aligned(b, 16) && (!c || focus_aligned(c, 16)) && (!d || focus_aligned(d, 16));
}

#define LN_V16_DISPATCH(CALL)                                        \
    do {                                                             \
        if (D <= 64) CALL(8, 1); else if (D <= 128) CALL(16, 1);     \
        else if (D <= 256) CALL(32, 1); else if (D <= 512) CALL(64, 1); \
        else CALL(64, 2);                                            \
    } while (0)

template <typename T>
int ln_fwd_launch(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int rows, int D,
                  float eps, int rpb, int64_t xbs, hipStream_t s) {
    const int nv = (D + 255) / 256;
    dim3 grid((rows + 3) / 4), blk(256);
#define LN_FWD(NV) hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), grid, blk, 0, s, (const T*)x, g, b, (T*)y, mean, rstd, rows, D, eps, rpb, xbs)
    if (nv <= 1) LN_FWD(1); else if (nv <= 2) LN_FWD(2); else if (nv <= 3) LN_FWD(3); else if (nv <= 4) LN_FWD(4);
    else if (nv <= 8) LN_FWD(8); else LN_FWD(16);
#undef LN_FWD
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <typename T>
int ln_bwd_launch(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, const void* dres,
                  void* dx, float* partial, int rows, int D, int nblk, int rpb, int64_t xbs, hipStream_t s) {
    const int nv = (D + 255) / 256;
    dim3 grid(nblk), blk(256);
#define LN_BWD(NV) hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), grid, blk, 0, s, (const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx, partial, rows, D, rpb, xbs)
    if (nv <= 1) LN_BWD(1); else if (nv <= 2) LN_BWD(2); else if (nv <= 3) LN_BWD(3); else if (nv <= 4) LN_BWD(4);
    else if (nv <= 8) LN_BWD(8); else LN_BWD(16);
#undef LN_BWD
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

static int ln_fwd_any(const void* x, int rpb, int64_t xbs, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int rows, int D, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return FOCUS_ERR_NULL;
    if (rows <= 0) return FOCUS_OK;
    if (D <= 0 || (D & 3) || D > MAXV * 256 || rpb <= 0 || (xbs & 3)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(x, 8) || !focus_aligned(y, 8) || !focus_aligned(gamma, 16) || !focus_aligned(beta, 16))
        return FOCUS_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (ln_v16_ok(x, y, gamma, beta, rows, D, xbs, dtype)) {
#define LNF(LPR, NV) do { \
        constexpr int per_blk = 4 * (64 / LPR) * (NV == 1 ? 4 : 2); \
        const int grid = (int)std::min<int64_t>(((int64_t)rows + per_blk - 1) / per_blk, 4096); \
        hipLaunchKernelGGL((ln_fwd_v16_kernel<LPR, NV>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, D, eps, rpb, xbs); } while (0)
        LN_V16_DISPATCH(LNF);
#undef LNF
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    return dtype == FOCUS_BF16 ? ln_fwd_launch<bf16_t>(x, gamma, beta, y, mean, rstd, rows, D, eps, rpb, xbs, s)
                               : ln_fwd_launch<float>(x, gamma, beta, y, mean, rstd, rows, D, eps, rpb, xbs, s);
}

extern "C" int focus_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, int rows, int D, float eps, int dtype, void* stream) {
    return ln_fwd_any(x, rows > 0 ? rows : 1, 0, gamma, beta, y, mean, rstd, rows, D, eps, dtype, stream);
}

extern "C" int focus_layernorm_fwd_blocks(const void* x, int rows_per_block, int64_t block_stride, const float* gamma,
                                          const float* beta, void* y, float* mean, float* rstd, int rows, int D, float eps,
                                          int dtype, void* stream) {
    return ln_fwd_any(x, rows_per_block, block_stride, gamma, beta, y, mean, rstd, rows, D, eps, dtype, stream);
}

extern "C" int focus_layernorm_bwd_blocks(int rows) {
    int b = (rows + 3) / 4;
    return b < 1 ? 1 : (b > 512 ? 512 : b);
}

static int ln_bwd_any(const void* dy, const void* x, int rpb, int64_t xbs, const float* gamma, const float* mean,
                      const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* partial, int rows,
                      int D, int dtype, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !partial) return FOCUS_ERR_NULL;
    if (!dgamma != !dbeta) return FOCUS_ERR_NULL;                   // both, or neither: the caller sums `partial` itself
    const bool finish = dgamma != nullptr;
    if (D <= 0 || (D & 3) || D > MAXV * 256 || rows <= 0 || rpb <= 0 || (xbs & 3)) return FOCUS_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = focus_layernorm_bwd_blocks(rows);
    if (ln_v16_ok(dy, x, dx, dres, rows, D, xbs, dtype) && focus_aligned(gamma, 16)) {
    // nblk is capped (the finish walks nblk partial rows): beyond 64 rows per wave the workgroups get 8 waves instead of 4,
    // so that a CU holds 16 waves' worth of loads in flight (frame tokens of STEVE: 131072 rows)
    const bool wide = (int64_t)rows >= (int64_t)nblk * 4 * 64;
#define LNB(LPR, NV) do { \
        if (wide) { \
            const size_t lds = (size_t)2 * 8 * (64 / LPR) * D * sizeof(float); \
            hipLaunchKernelGGL((ln_bwd_v16_kernel<LPR, NV, 8>), dim3(nblk), dim3(512), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, partial, rows, D, rpb, xbs); \
        } else { \
            const size_t lds = (size_t)2 * 4 * (64 / LPR) * D * sizeof(float); \
            hipLaunchKernelGGL((ln_bwd_v16_kernel<LPR, NV, 4>), dim3(nblk), dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, partial, rows, D, rpb, xbs); } } while (0)
        LN_V16_DISPATCH(LNB);
#undef LNB
        FOCUS_CHECK_LAUNCH();
        if (finish) {
            hipLaunchKernelGGL(ln_bwd_finish, dim3((D + 63) / 64, 2), dim3(1024), 0, s, partial, dgamma, dbeta, nblk, D);
            FOCUS_CHECK_LAUNCH();
        }
        return FOCUS_OK;
    }
    int rc = dtype == FOCUS_BF16 ? ln_bwd_launch<bf16_t>(dy, x, gamma, mean, rstd, dres, dx, partial, rows, D, nblk, rpb, xbs, s)
                                 : ln_bwd_launch<float>(dy, x, gamma, mean, rstd, dres, dx, partial, rows, D, nblk, rpb, xbs, s);
    if (rc) return rc;
    if (finish) {
        hipLaunchKernelGGL(ln_bwd_finish, dim3((D + 63) / 64, 2), dim3(1024), 0, s, partial, dgamma, dbeta, nblk, D);
        FOCUS_CHECK_LAUNCH();
    }
    return FOCUS_OK;
}

extern "C" int focus_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                   float* partial, int rows, int D, int dtype, void* stream) {
    return ln_bwd_any(dy, x, rows > 0 ? rows : 1, 0, gamma, mean, rstd, dres, dx, dgamma, dbeta, partial, rows, D, dtype, stream);
}

extern "C" int focus_layernorm_bwd_blocks_strided(const void* dy, const void* x, int rows_per_block, int64_t block_stride,
                                                  const float* gamma, const float* mean, const float* rstd, void* dx,
                                                  float* dgamma, float* dbeta, float* partial, int rows, int D, int dtype,
                                                  void* stream) {
    return ln_bwd_any(dy, x, rows_per_block, block_stride, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, partial, rows, D,
                      dtype, stream);
}
