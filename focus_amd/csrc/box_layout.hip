// box_layout.hip -- box "layout" of the ORViT motion stream (ORViT/utils.py:8-28 -> layout.py:28-63,
// 98-130, 205-237) in closed form:
//   out[n,y,x,:] = sum_o keep_o * vec[n,o,:] * w((lin_y - y0_o)/y1_o) * w((lin_x - x0_o)/x1_o)
//   w(u) = clamp(min(7u+1, 8-7u), 0, 1),  lin = linspace(0,1,H),  (x0,y0,x1,y1) = cxcywh->xyxy,
//   keep_o = any(xyxy_o != 0).  The division by x1,y1 (not width,height) is the reference's behaviour.
// Replaces a B x T python loop of ~12 tiny ops per frame by one launch; HBM-bound on the output write.
#include "focus_common.h"

namespace {

constexpr int MAXO = 16;

__device__ __forceinline__ float wfun(float u) { return fminf(fmaxf(fminf(7.f * u + 1.f, 8.f - 7.f * u), 0.f), 1.f); }

struct BoxW { float x0, y0, x1, y1; bool keep; };
__device__ __forceinline__ BoxW box_xyxy(const float* b) {
    BoxW r;
    r.x0 = b[0] - 0.5f * b[2]; r.y0 = b[1] - 0.5f * b[3];
    r.x1 = b[0] + 0.5f * b[2]; r.y1 = b[1] + 0.5f * b[3];
    r.keep = (r.x0 != 0.f) || (r.y0 != 0.f) || (r.x1 != 0.f) || (r.y1 != 0.f);
    return r;
}
__device__ __forceinline__ float lin(int i, int n) { return n > 1 ? (float)i / (float)(n - 1) : 0.f; }

// grid (H, NF): one block per output row of one frame.
template <typename T>
__global__ __launch_bounds__(256) void layout_fwd_kernel(const T* __restrict__ vecs, const float* __restrict__ boxes,
                                                         T* __restrict__ out, int O, int C, int H, int W) {
    __shared__ float wy[MAXO];
    __shared__ float wx[MAXO][64];
    const int n = blockIdx.y, y = blockIdx.x;
    for (int i = threadIdx.x; i < O * W; i += 256) {
        const int o = i / W, x = i % W;
        const BoxW b = box_xyxy(boxes + ((int64_t)n * O + o) * 4);
        wx[o][x] = b.keep ? wfun((lin(x, W) - b.x0) / b.x1) : 0.f;
        if (x == 0) wy[o] = b.keep ? wfun((lin(y, H) - b.y0) / b.y1) : 0.f;
    }
    __syncthreads();
    const int cq = C >> 2;
    for (int it = threadIdx.x; it < W * cq; it += 256) {
        const int x = it / cq, c = (it % cq) * 4;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < O; ++o) {
            const float w = wy[o] * wx[o][x];
            if (w != 0.f) {
                const f4 v = ld4<T>(vecs + ((int64_t)n * O + o) * C + c);
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        st4<T>(out + (((int64_t)n * H + y) * W + x) * C + c, acc);
    }
}

// grid (O, NF): dvec[n,o,:] = sum_{y,x} wy*wx*dout[n,y,x,:].  384 threads = up to 96 channel octets (16-byte loads) x 4
// row groups that meet in LDS.  (One thread per 4 channels walking all H*W cells by itself took 63 us for 19 MB.)
constexpr int LB_GROUPS = 4, LB_OCT = 96;
template <typename T>
__global__ __launch_bounds__(LB_GROUPS * LB_OCT) void layout_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ boxes,
                                                                     T* __restrict__ dvecs, int O, int C, int H, int W) {
    __shared__ float wy[64], wx[64];
    __shared__ float red[LB_GROUPS][LB_OCT * 8];
    const int n = blockIdx.y, o = blockIdx.x;
    const BoxW b = box_xyxy(boxes + ((int64_t)n * O + o) * 4);
    for (int i = threadIdx.x; i < H + W; i += LB_GROUPS * LB_OCT) {
        if (i < H) wy[i] = b.keep ? wfun((lin(i, H) - b.y0) / b.y1) : 0.f;
        else wx[i - H] = b.keep ? wfun((lin(i - H, W) - b.x0) / b.x1) : 0.f;
    }
    __syncthreads();
    const int oct = threadIdx.x % LB_OCT, grp = threadIdx.x / LB_OCT, noct = C >> 3;
    for (int c0 = 0; c0 < noct; c0 += LB_OCT) {                    // (one pass for C <= 768)
        const int c = (c0 + oct) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c0 + oct < noct) {
            // group grp takes the rows y = grp, grp + 4, ...; a row without weight is skipped by the whole workgroup's group
            // (the box is the same for all of it), a row with weight is read whole, its loads in flight together
            for (int y = grp; y < H; y += LB_GROUPS) {
                const float wrow = wy[y];
                if (wrow == 0.f) continue;
                const T* rowp = dout + (((int64_t)n * H + y) * W) * C + c;
#pragma unroll 7
                for (int x = 0; x < W; ++x) {
                    float v[8];
                    ld8<T>(rowp + (int64_t)x * C, v);
                    const float w = wrow * wx[x];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, v[e], acc[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[grp][oct * 8 + e] = acc[e];
        __syncthreads();
        if (grp == 0 && c0 + oct < noct) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = red[0][oct * 8 + e] + red[1][oct * 8 + e] + red[2][oct * 8 + e] + red[3][oct * 8 + e];
            st8<T>(dvecs + ((int64_t)n * O + o) * C + c, t);
        }
        __syncthreads();
    }
}

// any C % 4 == 0 / alignment: one thread per 4 channels (grid (O, NF))
template <typename T>
__global__ __launch_bounds__(256) void layout_bwd_scalar_kernel(const T* __restrict__ dout, const float* __restrict__ boxes,
                                                         T* __restrict__ dvecs, int O, int C, int H, int W) {
    __shared__ float wy[64], wx[64];
    const int n = blockIdx.y, o = blockIdx.x;
    const BoxW b = box_xyxy(boxes + ((int64_t)n * O + o) * 4);
    for (int i = threadIdx.x; i < H + W; i += 256) {
        if (i < H) wy[i] = b.keep ? wfun((lin(i, H) - b.y0) / b.y1) : 0.f;
        else wx[i - H] = b.keep ? wfun((lin(i - H, W) - b.x0) / b.x1) : 0.f;
    }
    __syncthreads();
    const int cq = C >> 2;
    for (int it = threadIdx.x; it < cq; it += 256) {
        const int c = it * 4;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < H; ++y) {
            if (wy[y] == 0.f) continue;
            for (int x = 0; x < W; ++x) {
                const float w = wy[y] * wx[x];
                if (w == 0.f) continue;
                const f4 v = ld4<T>(dout + (((int64_t)n * H + y) * W + x) * C + c);
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        st4<T>(dvecs + ((int64_t)n * O + o) * C + c, acc);
    }
}

}  // namespace

extern "C" int focus_box_layout_fwd(const void* vecs, const float* boxes, void* out, int NF, int O, int C, int H,
                                    int W, int dtype, void* stream) {
    if (!vecs || !boxes || !out) return FOCUS_ERR_NULL;
    if (NF <= 0) return FOCUS_OK;
    if ((C & 3) || O > MAXO || O < 0 || W > 64 || H > 64 || NF > 65535) return FOCUS_ERR_SHAPE;
    dim3 grid(H, NF);
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((layout_fwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)vecs,
                           boxes, (bf16_t*)out, O, C, H, W);
    else
        hipLaunchKernelGGL((layout_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)vecs,
                           boxes, (float*)out, O, C, H, W);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_box_layout_bwd(const void* dout, const float* boxes, void* dvecs, int NF, int O, int C, int H,
                                    int W, int dtype, void* stream) {
    if (!dout || !boxes || !dvecs) return FOCUS_ERR_NULL;
    if (NF <= 0 || O <= 0) return FOCUS_OK;
    if ((C & 3) || O > MAXO || W > 64 || H > 64 || NF > 65535) return FOCUS_ERR_SHAPE;
    dim3 grid(O, NF);
    const bool wide = (C & 7) == 0 && focus_aligned(dout, 16) && focus_aligned(dvecs, 16);
    if (wide && dtype == FOCUS_BF16)
        hipLaunchKernelGGL((layout_bwd_kernel<bf16_t>), grid, dim3(LB_GROUPS * LB_OCT), 0, (hipStream_t)stream,
                           (const bf16_t*)dout, boxes, (bf16_t*)dvecs, O, C, H, W);
    else if (wide)
        hipLaunchKernelGGL((layout_bwd_kernel<float>), grid, dim3(LB_GROUPS * LB_OCT), 0, (hipStream_t)stream,
                           (const float*)dout, boxes, (float*)dvecs, O, C, H, W);
    else if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((layout_bwd_scalar_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                           boxes, (bf16_t*)dvecs, O, C, H, W);
    else
        hipLaunchKernelGGL((layout_bwd_scalar_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dout,
                           boxes, (float*)dvecs, O, C, H, W);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
