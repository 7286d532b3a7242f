// roi_align.hip -- RoIAlign over channels-last patch-token feature maps (ORViT/utils.py:58-75).
// Geometry follows oracle/roi_align_ref.c operation by operation in IEEE fp32 WITHOUT fma contraction
// (__fmul_rn/__fadd_rn/...), so sampling-grid sizes and neighbour indices are bit-exact with the oracle.
// HBM-bound: per output cell and channel, 4 gathered reads (L2-resident: a 14x14x768 map is 300 KB)
// and one write.
#include "focus_common.h"
#include <cstdlib>

namespace {

struct Geom { float y1, x1, bin_h, bin_w; int grid_h, grid_w; float count; };
struct Nbr { int y_low, x_low, y_high, x_high; float w1, w2, w3, w4; };

__device__ __forceinline__ Geom roi_geometry(const float* roi, float scale, int PH, int PW, int sampling_ratio,
                                             int aligned) {
    Geom g;
    const float off = aligned ? 0.5f : 0.0f;
    g.x1 = __fsub_rn(__fmul_rn(roi[0], scale), off);
    g.y1 = __fsub_rn(__fmul_rn(roi[1], scale), off);
    const float x2 = __fsub_rn(__fmul_rn(roi[2], scale), off);
    const float y2 = __fsub_rn(__fmul_rn(roi[3], scale), off);
    float rw = __fsub_rn(x2, g.x1), rh = __fsub_rn(y2, g.y1);
    if (!aligned) { if (rw < 1.0f) rw = 1.0f; if (rh < 1.0f) rh = 1.0f; }
    g.bin_h = __fdiv_rn(rh, (float)PH);
    g.bin_w = __fdiv_rn(rw, (float)PW);
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(__fdiv_rn(rh, (float)PH));
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(__fdiv_rn(rw, (float)PW));
    const int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}
__device__ __forceinline__ float sample_coord(float start, int p, float bin, int i, int grid) {
    // start + p*bin + (i + 0.5)*bin/grid, evaluated left to right as in the oracle
    return __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)),
                     __fdiv_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), bin), (float)grid));
}
__device__ __forceinline__ Nbr locate(float y, float x, int H, int W) {
    Nbr n;
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
        n.y_low = n.x_low = n.y_high = n.x_high = -1;
        n.w1 = n.w2 = n.w3 = n.w4 = 0.f;
        return n;
    }
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    n.y_low = (int)y;
    n.x_low = (int)x;
    if (n.y_low >= H - 1) { n.y_high = n.y_low = H - 1; y = (float)n.y_low; } else n.y_high = n.y_low + 1;
    if (n.x_low >= W - 1) { n.x_high = n.x_low = W - 1; x = (float)n.x_low; } else n.x_high = n.x_low + 1;
    const float ly = __fsub_rn(y, (float)n.y_low), lx = __fsub_rn(x, (float)n.x_low);
    const float hy = __fsub_rn(1.0f, ly), hx = __fsub_rn(1.0f, lx);
    n.w1 = __fmul_rn(hy, hx); n.w2 = __fmul_rn(hy, lx); n.w3 = __fmul_rn(ly, hx); n.w4 = __fmul_rn(ly, lx);
    return n;
}

// grid: (PH, K); block 256 threads sweep (pw, channel-quad) of one output row of one RoI.
template <typename T>
__global__ __launch_bounds__(256) void roi_fwd_kernel(const T* __restrict__ feat, int64_t img_stride, int ipb,
                                                      int64_t batch_stride, const float* __restrict__ rois,
                                                      const int32_t* __restrict__ roi_img, T* __restrict__ out, int C,
                                                      int H, int W, int PH, int PW, float scale, int sr, int aligned,
                                                      int relu) {
    const int k = blockIdx.y, ph = blockIdx.x;
    const Geom g = roi_geometry(rois + 4 * k, scale, PH, PW, sr, aligned);
    const int ri = roi_img[k];
    const T* img = feat + (int64_t)(ri / ipb) * batch_stride + (int64_t)(ri % ipb) * img_stride;
    const int cq = C >> 2;
    for (int it = threadIdx.x; it < PW * cq; it += 256) {
        const int pw = it / cq, c = (it % cq) * 4;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = sample_coord(g.y1, ph, g.bin_h, iy, g.grid_h);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = sample_coord(g.x1, pw, g.bin_w, ix, g.grid_w);
                const Nbr n = locate(y, x, H, W);
                if (n.y_low < 0) continue;
                const f4 a = ld4<T>(img + (int64_t)(n.y_low * W + n.x_low) * C + c);
                const f4 b = ld4<T>(img + (int64_t)(n.y_low * W + n.x_high) * C + c);
                const f4 cc = ld4<T>(img + (int64_t)(n.y_high * W + n.x_low) * C + c);
                const f4 dd = ld4<T>(img + (int64_t)(n.y_high * W + n.x_high) * C + c);
                // same association as the oracle: acc += w1*a + w2*b + w3*c + w4*d
                acc.x += __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(n.w1, a.x), __fmul_rn(n.w2, b.x)), __fmul_rn(n.w3, cc.x)), __fmul_rn(n.w4, dd.x));
                acc.y += __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(n.w1, a.y), __fmul_rn(n.w2, b.y)), __fmul_rn(n.w3, cc.y)), __fmul_rn(n.w4, dd.y));
                acc.z += __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(n.w1, a.z), __fmul_rn(n.w2, b.z)), __fmul_rn(n.w3, cc.z)), __fmul_rn(n.w4, dd.z));
                acc.w += __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(n.w1, a.w), __fmul_rn(n.w2, b.w)), __fmul_rn(n.w3, cc.w)), __fmul_rn(n.w4, dd.w));
            }
        }
        acc.x = __fdiv_rn(acc.x, g.count); acc.y = __fdiv_rn(acc.y, g.count);
        acc.z = __fdiv_rn(acc.z, g.count); acc.w = __fdiv_rn(acc.w, g.count);
        if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        st4<T>(out + ((int64_t)k * PH * PW + ph * PW + pw) * C + c, acc);
    }
}

// One lane per channel: every atomic wave-instruction adds 256 contiguous bytes of one feature cell (the shape the
// memory-side float atomics run fastest at -- MI355X_MICROARCH.md "Global float atomics").
template <typename T>
__global__ __launch_bounds__(256) void roi_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ rois,
                                                      const int32_t* __restrict__ roi_img, float* __restrict__ dfeat,
                                                      int C, int H, int W, int PH, int PW, float scale, int sr,
                                                      int aligned) {
    const int k = blockIdx.y, ph = blockIdx.x;
    const Geom g = roi_geometry(rois + 4 * k, scale, PH, PW, sr, aligned);
    float* img = dfeat + (int64_t)roi_img[k] * H * W * C;
    for (int pw = 0; pw < PW; ++pw) {
        const T* drow = dout + ((int64_t)k * PH * PW + ph * PW + pw) * C;
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = sample_coord(g.y1, ph, g.bin_h, iy, g.grid_h);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = sample_coord(g.x1, pw, g.bin_w, ix, g.grid_w);
                const Nbr n = locate(y, x, H, W);           // block-uniform
                if (n.y_low < 0) continue;
                float* p1 = img + (int64_t)(n.y_low * W + n.x_low) * C;
                float* p2 = img + (int64_t)(n.y_low * W + n.x_high) * C;
                float* p3 = img + (int64_t)(n.y_high * W + n.x_low) * C;
                float* p4 = img + (int64_t)(n.y_high * W + n.x_high) * C;
                for (int c = threadIdx.x; c < C; c += 256) {
                    const float gv = ld<T>(drow + c) / g.count;
                    if (n.w1 != 0.f) atomicAdd(p1 + c, gv * n.w1);
                    if (n.w2 != 0.f) atomicAdd(p2 + c, gv * n.w2);
                    if (n.w3 != 0.f) atomicAdd(p3 + c, gv * n.w3);
                    if (n.w4 != 0.f) atomicAdd(p4 + c, gv * n.w4);
                }
            }
        }
    }
}

// Backward without atomics, using the separability of RoIAlign: a sampling point's bilinear weight on cell (y, x) is
// wy(y) * wx(x), its validity test is (y valid) && (x valid), and the points of a RoI form a grid, so for one RoI
//     dfeat[y][x][c] += (1/count) * sum_ph sum_pw  Ay[y][ph] * dout[ph][pw][c] * Ax[x][pw]
// with Ay[y][ph] = sum over the bin row's sampling points of wy (a [H x PH] matrix, likewise Ax [W x PW]).
// One workgroup = (image, 64 channels); thread = (channel, column x) keeps its H output cells in registers over all
// RoIs of the image: T1[ph] = sum_pw dout[ph][pw] * Ax[x][pw], then acc[y] += sum_ph Ay[y][ph] * T1[ph].
// No atomics, no zero fill, dout read once, dfeat written once in the feature dtype.  (The scatter form needs 4
// float atomics per sampling point and channel: 154 M global atomics per call at the bench shape, 474 us; LDS float
// atomics were slower still -- ds_add_f32 retires about one lane per clock.)
// Two instances: <64 channels, 16, 16> for maps and crops up to 16 x 16 (the 14 x 14 of the 224 crops; block = 64 x W <= 1024
// threads), <32 channels, 24, 24> for up to 24 x 24 (the 21 x 21 of the HR 336 crops, which the first cannot hold: the
// atomic fallback took 560 us per call there).  CS = channels per workgroup, HM / PM = register-array bounds for H, PH / PW.

// 1-D half of locate(): v -> (low, high, weight of low, weight of high); low < 0 when the coordinate is out of range
__device__ __forceinline__ void locate1(float v, int L, int* low, int* high, float* w_low, float* w_high) {
    if (v < -1.0f || v > (float)L) { *low = -1; *high = -1; *w_low = 0.f; *w_high = 0.f; return; }
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= L - 1) { hi = lo = L - 1; v = (float)lo; } else hi = lo + 1;
    const float l = __fsub_rn(v, (float)lo);
    *low = lo; *high = hi; *w_low = __fsub_rn(1.0f, l); *w_high = l;
}

template <typename T, int CS, int HM, int PM>
__global__ __launch_bounds__(CS * HM) void roi_bwd_sep_kernel(const T* __restrict__ dout, const T* __restrict__ relu_out,
                                                           const float* __restrict__ rois,
                                                           const int32_t* __restrict__ roi_img, T* __restrict__ dfeat,
                                                           int64_t img_stride, int ipb, int64_t batch_stride,
                                                           int C, int H, int W, int K, int PH, int PW, float scale,
                                                           int sr, int aligned) {
    extern __shared__ __attribute__((aligned(16))) float gst[];   // [bins][CS]: dout / count of the current RoI
    __shared__ __attribute__((aligned(16))) float Ay[HM][PM], Ax[HM][PM];
    __shared__ int match[1024], wcnt[16], nmatch_s;
    const int img = blockIdx.y, c0 = blockIdx.x * CS;
    const int ch = threadIdx.x % CS, x = threadIdx.x / CS;                 // x < W (blockDim = CS * W)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nwv = (blockDim.x + 63) >> 6, nx = blockDim.x / CS, bins = PH * PW;
    float acc[HM];
#pragma unroll
    for (int y = 0; y < HM; ++y) acc[y] = 0.f;
    for (int kb = 0; kb < K; kb += blockDim.x) {
        // the RoIs of this image, in index order
        const int kk = kb + threadIdx.x;
        const bool m = kk < K && roi_img[kk] == img;
        const unsigned long long bal = __ballot(m);
        __syncthreads();
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int j = 0; j < nwv; ++j) { if (j < wv) off += wcnt[j]; tot += wcnt[j]; }
        if (m) match[off + __popcll(bal & ((1ull << lane) - 1ull))] = kk;
        if (threadIdx.x == 0) nmatch_s = tot;
        __syncthreads();
        const int nmatch = nmatch_s;
        for (int mi = 0; mi < nmatch; ++mi) {
            const int k = match[mi];
            const Geom g = roi_geometry(rois + 4 * k, scale, PH, PW, sr, aligned);
            if (g.grid_h <= 0 || g.grid_w <= 0) continue;                  // block-uniform
            __syncthreads();                                               // previous RoI's gst / Ay / Ax are free
            constexpr int NB = PM;                      // bins per thread held in registers (bins <= NB * columns)
            if (bins <= NB * nx) {
                // all loads of this thread issued together (unrolled, clamped addresses instead of a branch): as a rolled
                // "load, store to LDS" loop every iteration waited for its own load -- 14 serial round trips per RoI, the
                // whole launch ran at 3 % of the HBM rate (215 us for 48 MB)
                float gv[NB], rv[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int bin = min(x + i * nx, bins - 1);
                    const int64_t o = ((int64_t)k * bins + bin) * C + c0 + ch;
                    gv[i] = ld<T>(dout + o);
                    rv[i] = relu_out ? ld<T>(relu_out + o) : 1.f;
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int bin = x + i * nx;
                    if (bin < bins) gst[bin * CS + ch] = rv[i] > 0.f ? gv[i] / g.count : 0.f;
                }
            } else {
                for (int bin = x; bin < bins; bin += nx) {
                    const int64_t o = ((int64_t)k * bins + bin) * C + c0 + ch;
                    float gv = ld<T>(dout + o) / g.count;
                    if (relu_out && !(ld<T>(relu_out + o) > 0.f)) gv = 0.f;     // fused ReLU: the saved output is the mask
                    gst[bin * CS + ch] = gv;
                }
            }
            for (int i = threadIdx.x; i < HM * PM; i += blockDim.x) { (&Ay[0][0])[i] = 0.f; (&Ax[0][0])[i] = 0.f; }
            __syncthreads();
            // one thread per bin row / bin column: no two threads write the same Ay / Ax entry
            if (threadIdx.x < PH) {
                const int ph = threadIdx.x;
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    int lo, hi; float wl, wh;
                    locate1(sample_coord(g.y1, ph, g.bin_h, iy, g.grid_h), H, &lo, &hi, &wl, &wh);
                    if (lo >= 0) { Ay[lo][ph] += wl; Ay[hi][ph] += wh; }
                }
            } else if (threadIdx.x >= 64 && threadIdx.x < 64 + PW) {
                const int pw = threadIdx.x - 64;
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    int lo, hi; float wl, wh;
                    locate1(sample_coord(g.x1, pw, g.bin_w, ix, g.grid_w), W, &lo, &hi, &wl, &wh);
                    if (lo >= 0) { Ax[lo][pw] += wl; Ax[hi][pw] += wh; }
                }
            }
            __syncthreads();
            // this thread's row of Ax in registers (ds_read_b128s instead of PH*PW broadcast reads)
            float ax[PM];
#pragma unroll
            for (int q4 = 0; q4 < PM / 4; ++q4) {
                const float4 v = *reinterpret_cast<const float4*>(&Ax[x][q4 * 4]);
                ax[q4 * 4 + 0] = v.x; ax[q4 * 4 + 1] = v.y; ax[q4 * 4 + 2] = v.z; ax[q4 * 4 + 3] = v.w;
            }
            float t1[PM];
#pragma unroll
            for (int ph = 0; ph < PM; ++ph) {
                float t = 0.f;
                if (ph < PH) {
#pragma unroll
                    for (int pw = 0; pw < PM; ++pw)
                        if (pw < PW) t = fmaf(gst[(ph * PW + pw) * CS + ch], ax[pw], t);
                }
                t1[ph] = t;
            }
#pragma unroll
            for (int y = 0; y < HM; ++y) {
                if (y < H) {
                    float t = 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < PM / 4; ++q4) {
                        const float4 a = *reinterpret_cast<const float4*>(&Ay[y][q4 * 4]);   // zero beyond PH
                        t = fmaf(a.x, t1[q4 * 4 + 0], t); t = fmaf(a.y, t1[q4 * 4 + 1], t);
                        t = fmaf(a.z, t1[q4 * 4 + 2], t); t = fmaf(a.w, t1[q4 * 4 + 3], t);
                    }
                    acc[y] += t;
                }
            }
        }
    }
#pragma unroll
    for (int y = 0; y < HM; ++y)
        if (y < H)
            st<T>(dfeat + (int64_t)(img / ipb) * batch_stride + (int64_t)(img % ipb) * img_stride + (int64_t)(y * W + x) * C + c0 + ch,
                  acc[y]);
}

// which separable instance takes the shape: 1 = <64, 16, 16>, 2 = <32, 24, 24>, 0 = none (atomic kernel + cast)
static int roi_sep_ok(int C, int H, int W, int PH, int PW) {
    static const int mode = getenv("FOCUS_ROI_BWD") ? atoi(getenv("FOCUS_ROI_BWD")) : 1;
    if (mode == 0 || W < 2) return 0;
    if (H <= 16 && W <= 16 && PH <= 16 && PW <= 16 && PH * PW * 64 * 4 <= 56 * 1024 && C % 64 == 0) return 1;
    if (H <= 24 && W <= 24 && PH <= 24 && PW <= 24 && PH * PW * 32 * 4 <= 60 * 1024 && C % 32 == 0) return 2;
    return 0;
}

__global__ void roi_indices_kernel(const float* __restrict__ rois, int32_t* __restrict__ grid,
                                   int32_t* __restrict__ nbr, int H, int W, int K, int PH, int PW, float scale,
                                   int sr, int aligned) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * PH * PW) return;
    const int pw = i % PW, ph = (i / PW) % PH, k = i / (PW * PH);
    const Geom g = roi_geometry(rois + 4 * k, scale, PH, PW, sr, aligned);
    if (ph == 0 && pw == 0) { grid[2 * k] = g.grid_h; grid[2 * k + 1] = g.grid_w; }
    int32_t* o = nbr + (int64_t)i * 4;
    if (g.grid_h <= 0 || g.grid_w <= 0) { o[0] = o[1] = o[2] = o[3] = -1; return; }
    const Nbr n = locate(sample_coord(g.y1, ph, g.bin_h, 0, g.grid_h), sample_coord(g.x1, pw, g.bin_w, 0, g.grid_w),
                         H, W);
    o[0] = n.y_low; o[1] = n.x_low; o[2] = n.y_high; o[3] = n.x_high;
}

}  // namespace

extern "C" int focus_roi_align_fwd(const void* feat, int64_t img_stride, int imgs_per_batch, int64_t batch_stride,
                                   const float* rois, const int32_t* roi_img, void* out, int NI, int C, int H, int W,
                                   int K, int PH, int PW, float scale, int sr, int aligned, int relu, int dtype,
                                   void* stream) {
    (void)NI;
    if (!feat || !rois || !roi_img || !out) return FOCUS_ERR_NULL;
    if (K <= 0) return FOCUS_OK;
    if (imgs_per_batch <= 0 || (img_stride & 3) || (batch_stride & 3)) return FOCUS_ERR_SHAPE;
    if ((C & 3) || PH <= 0 || PW <= 0 || K > 65535 * 1024) return FOCUS_ERR_SHAPE;
    if (K > 65535) return FOCUS_ERR_SHAPE;
    dim3 grid(PH, K);
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((roi_fwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat,
                           img_stride, imgs_per_batch, batch_stride, rois, roi_img, (bf16_t*)out, C, H, W, PH, PW, scale, sr, aligned, relu);
    else
        hipLaunchKernelGGL((roi_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)feat,
                           img_stride, imgs_per_batch, batch_stride, rois, roi_img, (float*)out, C, H, W, PH, PW, scale, sr, aligned, relu);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" size_t focus_roi_align_bwd_workspace_bytes(int NI, int C, int H, int W, int PH, int PW) {
    if (roi_sep_ok(C, H, W, PH, PW)) return 0;
    return (size_t)NI * H * W * C * sizeof(float);
}

extern "C" int focus_roi_align_bwd(const void* dout, const void* relu_out, const float* rois, const int32_t* roi_img, void* dfeat,
                                   int64_t img_stride, int imgs_per_batch, int64_t batch_stride, void* ws,
                                   size_t ws_bytes, int NI, int C, int H, int W, int K, int PH, int PW, float scale,
                                   int sr, int aligned, int dtype, void* stream) {
    if (!dout || !rois || !roi_img || !dfeat) return FOCUS_ERR_NULL;
    if ((C & 3) || PH <= 0 || PW <= 0 || K > 65535 || NI <= 0 || NI > 65535 || imgs_per_batch <= 0) return FOCUS_ERR_SHAPE;
    const bool dense = img_stride == (int64_t)H * W * C && batch_stride == img_stride * imgs_per_batch;
    if (!dense && !roi_sep_ok(C, H, W, PH, PW)) return FOCUS_ERR_SHAPE;   // the atomic path accumulates into a dense map
    if (relu_out && !roi_sep_ok(C, H, W, PH, PW)) return FOCUS_ERR_SHAPE; // the fused-ReLU mask is a separable-path feature
    hipStream_t s = (hipStream_t)stream;
    const int sep = roi_sep_ok(C, H, W, PH, PW);
    if (sep) {
        const int cs = sep == 1 ? 64 : 32;
        dim3 grid(C / cs, NI), blk(cs * W);
        const size_t lds = (size_t)PH * PW * cs * sizeof(float);
#define ROI_SEP(T, CS, HM) do { \
            auto k = roi_bwd_sep_kernel<T, CS, HM, HM>; \
            if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return FOCUS_ERR_LAUNCH; \
            hipLaunchKernelGGL(k, grid, blk, lds, s, (const T*)dout, (const T*)relu_out, rois, roi_img, (T*)dfeat, img_stride, \
                               imgs_per_batch, batch_stride, C, H, W, K, PH, PW, scale, sr, aligned); } while (0)
        if (dtype == FOCUS_BF16) { if (sep == 1) ROI_SEP(bf16_t, 64, 16); else ROI_SEP(bf16_t, 32, 24); }
        else { if (sep == 1) ROI_SEP(float, 64, 16); else ROI_SEP(float, 32, 24); }
#undef ROI_SEP
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    // other shapes: fp32 atomics into a zeroed scratch map (or into dfeat itself when it is fp32), then one cast
    const size_t n = (size_t)NI * H * W * C;
    float* acc = dtype == FOCUS_F32 ? static_cast<float*>(dfeat) : static_cast<float*>(ws);
    if (dtype != FOCUS_F32 && (!ws || ws_bytes < n * sizeof(float))) return FOCUS_ERR_WORKSPACE;
    if (hipMemsetAsync(acc, 0, n * sizeof(float), s) != hipSuccess) return FOCUS_ERR_LAUNCH;
    if (K > 0) {
        dim3 grid(PH, K);
        if (dtype == FOCUS_BF16)
            hipLaunchKernelGGL((roi_bwd_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dout, rois, roi_img, acc, C,
                               H, W, PH, PW, scale, sr, aligned);
        else
            hipLaunchKernelGGL((roi_bwd_kernel<float>), grid, dim3(256), 0, s, (const float*)dout, rois, roi_img, acc, C, H,
                               W, PH, PW, scale, sr, aligned);
        FOCUS_CHECK_LAUNCH();
    }
    if (dtype != FOCUS_F32) return focus_cast(acc, FOCUS_F32, dfeat, dtype, (int64_t)n, stream);
    return FOCUS_OK;
}

extern "C" int focus_roi_align_indices(const float* rois, int32_t* grid, int32_t* nbr, int H, int W, int K, int PH,
                                       int PW, float scale, int sr, int aligned, void* stream) {
    if (!rois || !grid || !nbr) return FOCUS_ERR_NULL;
    if (K <= 0) return FOCUS_OK;
    const int n = K * PH * PW;
    hipLaunchKernelGGL(roi_indices_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rois, grid, nbr,
                       H, W, K, PH, PW, scale, sr, aligned);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
