// optim.hip -- the optimizer step of train_net.py:108-120 as two launches over ALL parameters of a model:
//   adamw_norm_kernel    sum of squares of every gradient (1024 block partials, deterministic order) + step counters;
//   adamw_update_kernel  clip coefficient from the partials (clip_grad_norm_, train_net.py:112-117), AdamW update of
//                        (param, exp_avg, exp_avg_sq) (torch.optim.AdamW, optimizer.py:137-145) and, in the same pass,
//                        the bf16 working copies of the updated weights (row-major for the forward GEMMs, transposed
//                        for the dX GEMMs: what autocast's per-use casts are in the reference, train_net.py:84).
// One read of grad / param / m / v and one write of param / m / v (+ the clipped grad, + 2 x 2 B of shadows) per
// element: ~30 B per parameter => 4.4 GB for the 147.5 M parameters of ORViT-MF, HBM-bound.
// Replaces torch's multi-tensor clip (20 launches), fused AdamW (9 launches) and the separate shadow refresh.
#include "focus_common.h"

namespace {

constexpr int NORM_BLOCKS = 1024;

// unit u of item `it`: a 64x64 tile (tiles mode: rows, cols multiples of 4 and a transposed shadow may be asked
// for) or a flat chunk of 4096 elements
__device__ __forceinline__ int find_item(const focus_adamw_item* __restrict__ items, int n_items, int unit) {
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].unit0 <= unit) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ bool tiled(const focus_adamw_item& it) { return it.tile_mode != 0; }

__global__ __launch_bounds__(256) void adamw_norm_kernel(const focus_adamw_item* __restrict__ items, float* const* __restrict__ grads,
                                                         int n_items, int n_units, float* __restrict__ steps,
                                                         float* __restrict__ partial) {
    __shared__ float red[16];
    float s = 0.f;
    const int t = threadIdx.x;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int ii = find_item(items, n_items, u);
        const focus_adamw_item it = items[ii];
        const int local = u - it.unit0;
        const float* __restrict__ itg = grads[ii];
        if (local == 0 && t == 0) steps[ii] += 1.0f;              // one writer per item; the update kernel reads it
        if (tiled(it)) {
            const int tiles_c = (it.cols + 63) >> 6;
            const int tr = local / tiles_c, tc = local - tr * tiles_c;
            const int lr = t >> 4, lc = (t & 15) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = tr * 64 + lr + 16 * k, c = tc * 64 + lc;
                if (r < it.rows && c < it.cols) {
                    const float4 g = *reinterpret_cast<const float4*>(itg + (int64_t)r * it.cols + c);
                    s += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
                }
            }
        } else {
            const int64_t n = (int64_t)it.rows * it.cols, base = (int64_t)local * 4096;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t i = base + k * 256 + t;
                if (i < n) { const float g = itg[i]; s += g * g; }
            }
        }
    }
    const float tot = block_sum(s, red);
    if (t == 0) partial[blockIdx.x] = tot;
}

// betas arrive as doubles: torch forms 1 - beta and 1 - beta^step in double precision before rounding to fp32
// (1.0f - 0.999f is 4.7e-5 away from (float)(1.0 - 0.999))
struct Hyper { double beta1, beta2; float eps, max_norm; };

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, float coef, float lr, float wd, float omb1, float b2,
                                      float omb2, float eps, float step_size, float inv_bc2_sqrt) {
    g *= coef;
    p -= lr * wd * p;
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_update_kernel(const focus_adamw_item* __restrict__ items, float* const* __restrict__ grads,
                                                           int n_items, const float* __restrict__ groups, const float* __restrict__ steps,
                                                           const float* __restrict__ partial, float* __restrict__ norm_out,
                                                           Hyper h, int write_grad) {
    __shared__ float red[16];
    __shared__ bf16_t tile[64][68];
    const int t = threadIdx.x;
    // total gradient norm: every block sums the same 1024 partials in the same order (4 KB from L2)
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NORM_BLOCKS / 256; ++k) s += partial[k * 256 + t];
    const float total = sqrtf(block_sum(s, red));
    float coef = 1.0f;
    if (h.max_norm > 0.f) coef = fminf(h.max_norm / (total + 1e-6f), 1.0f);
    if (blockIdx.x == 0 && t == 0 && norm_out) *norm_out = total;
    const int ii = find_item(items, n_items, blockIdx.x);
    const focus_adamw_item it = items[ii];
    float* __restrict__ itg = grads[ii];
    const int local = blockIdx.x - it.unit0;
    const float lr = groups[2 * it.group], wd = groups[2 * it.group + 1];
    const float step = steps[ii];
    const double bc1 = 1.0 - pow(h.beta1, (double)step), bc2 = 1.0 - pow(h.beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const float omb1 = (float)(1.0 - h.beta1), omb2 = (float)(1.0 - h.beta2), b2f = (float)h.beta2;
    const bool wg = write_grad && coef < 1.0f;
    if (tiled(it)) {
        const int tiles_c = (it.cols + 63) >> 6;
        const int tr = local / tiles_c, tc = local - tr * tiles_c;
        const int lr_ = t >> 4, lc = (t & 15) * 4;
        bf16_t* dst = static_cast<bf16_t*>(it.dst);
        bf16_t* dstT = static_cast<bf16_t*>(it.dstT);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = tr * 64 + lr_ + 16 * k, c = tc * 64 + lc;
            uint2 o = make_uint2(0, 0);
            if (r < it.rows && c < it.cols) {
                const int64_t off = (int64_t)r * it.cols + c;
                float4 p = *reinterpret_cast<const float4*>(it.p + off);
                float4 g = *reinterpret_cast<const float4*>(itg + off);
                float4 m = *reinterpret_cast<const float4*>(it.m + off);
                float4 v = *reinterpret_cast<const float4*>(it.v + off);
                adam1(p.x, g.x, m.x, v.x, coef, lr, wd, omb1, b2f, omb2, h.eps, step_size, inv_bc2_sqrt);
                adam1(p.y, g.y, m.y, v.y, coef, lr, wd, omb1, b2f, omb2, h.eps, step_size, inv_bc2_sqrt);
                adam1(p.z, g.z, m.z, v.z, coef, lr, wd, omb1, b2f, omb2, h.eps, step_size, inv_bc2_sqrt);
                adam1(p.w, g.w, m.w, v.w, coef, lr, wd, omb1, b2f, omb2, h.eps, step_size, inv_bc2_sqrt);
                *reinterpret_cast<float4*>(it.p + off) = p;
                *reinterpret_cast<float4*>(it.m + off) = m;
                *reinterpret_cast<float4*>(it.v + off) = v;
                if (wg) *reinterpret_cast<float4*>(itg + off) = g;
                o.x = (uint32_t)f32_to_bf16(p.x) | ((uint32_t)f32_to_bf16(p.y) << 16);
                o.y = (uint32_t)f32_to_bf16(p.z) | ((uint32_t)f32_to_bf16(p.w) << 16);
                if (dst) *reinterpret_cast<uint2*>(dst + off) = o;
            }
            if (dstT) *reinterpret_cast<uint2*>(&tile[lr_ + 16 * k][lc]) = o;
        }
        if (!dstT) return;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oc = tc * 64 + lr_ + 16 * k, orr = tr * 64 + lc;      // output row = source column
            if (oc < it.cols && orr < it.rows) {
                uint2 o;
                o.x = (uint32_t)tile[lc + 0][lr_ + 16 * k] | ((uint32_t)tile[lc + 1][lr_ + 16 * k] << 16);
                o.y = (uint32_t)tile[lc + 2][lr_ + 16 * k] | ((uint32_t)tile[lc + 3][lr_ + 16 * k] << 16);
                *reinterpret_cast<uint2*>(dstT + (int64_t)oc * it.rows + orr) = o;
            }
        }
    } else {
        const int64_t n = (int64_t)it.rows * it.cols, base = (int64_t)local * 4096;
        bf16_t* dst = static_cast<bf16_t*>(it.dst);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t i = base + k * 256 + t;
            if (i < n) {
                float p = it.p[i], g = itg[i], m = it.m[i], v = it.v[i];
                adam1(p, g, m, v, coef, lr, wd, omb1, b2f, omb2, h.eps, step_size, inv_bc2_sqrt);
                it.p[i] = p; it.m[i] = m; it.v[i] = v;
                if (wg) itg[i] = g;
                if (dst) dst[i] = f32_to_bf16(p);
            }
        }
    }
}

}  // namespace

extern "C" int focus_adamw_units(int rows, int cols, int tile_mode) {
    if (rows <= 0 || cols <= 0) return 0;
    if (tile_mode) return ((rows + 63) / 64) * ((cols + 63) / 64);
    return (int)(((int64_t)rows * cols + 4095) / 4096);
}

extern "C" size_t focus_adamw_workspace_bytes(void) { return NORM_BLOCKS * sizeof(float); }

extern "C" int focus_adamw_step(const focus_adamw_item* items, float* const* grads, int n_items, int n_units, const float* groups,
                                float* steps,
                                void* workspace, size_t workspace_bytes, float* total_norm, double beta1, double beta2, float eps,
                                float max_norm, int write_clipped_grads, void* stream) {
    if (!items || !grads || !groups || !steps || !workspace) return FOCUS_ERR_NULL;
    if (n_items <= 0 || n_units <= 0) return FOCUS_OK;
    if (workspace_bytes < NORM_BLOCKS * sizeof(float)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(adamw_norm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, items, grads, n_items, n_units, steps, partial);
    FOCUS_CHECK_LAUNCH();
    Hyper h = {beta1, beta2, eps, max_norm};
    hipLaunchKernelGGL(adamw_update_kernel, dim3(n_units), dim3(256), 0, s, items, grads, n_items, groups, steps, partial, total_norm, h,
                       write_clipped_grads);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
