/*
 * oracle/roi_align_ref.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement (plain C, fp32) of the RoIAlign arithmetic that the reference reaches through
 *   slowfast/models/ORViT/utils.py:64-71   torchvision.ops.roi_align(features, list(boxes), (H,W),
 *                                           spatial_scale, sampling_ratio=-1, aligned=True)
 * The algorithm lives in the third-party package torchvision (setup.py:25 "torchvision>=0.4.2",
 * no pinned version, sources absent from /root/reference and from this image).  This file restates
 * torchvision's published CPU algorithm (roi_align_kernel.cpp: pre-computed bilinear neighbours,
 * adaptive sampling grid ceil(roi/pool) when sampling_ratio<=0, half-pixel shift when aligned).
 * PARITY UNPINNED: the reference holds no test, fixture or golden vector for this call, and the
 * library itself cannot be imported here; the known answers in tests/test_oracle_roi_align.py
 * (identity resample, zero box, single-cell box) are hand-derived from the published algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int y_low, x_low, y_high, x_high;
    float w1, w2, w3, w4;
} neighbour_t;

/* One bilinear sample point -> its four integer neighbours and weights.
 * Out-of-range points (beyond one cell outside the map) contribute nothing: indices -1, weights 0. */
static neighbour_t locate(float y, float x, int H, int W) {
    neighbour_t n;
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
        n.y_low = n.x_low = n.y_high = n.x_high = -1;
        n.w1 = n.w2 = n.w3 = n.w4 = 0.0f;
        return n;
    }
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    n.y_low = (int)y;
    n.x_low = (int)x;
    if (n.y_low >= H - 1) { n.y_high = n.y_low = H - 1; y = (float)n.y_low; }
    else n.y_high = n.y_low + 1;
    if (n.x_low >= W - 1) { n.x_high = n.x_low = W - 1; x = (float)n.x_low; }
    else n.x_high = n.x_low + 1;
    float ly = y - (float)n.y_low, lx = x - (float)n.x_low;
    float hy = 1.0f - ly, hx = 1.0f - lx;
    n.w1 = hy * hx; n.w2 = hy * lx; n.w3 = ly * hx; n.w4 = ly * lx;
    return n;
}

typedef struct {
    float y1, x1, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
} roi_geom_t;

static roi_geom_t geometry(const float* roi /* x1,y1,x2,y2 */, float scale, int PH, int PW,
                           int sampling_ratio, int aligned) {
    roi_geom_t g;
    float off = aligned ? 0.5f : 0.0f;
    g.x1 = roi[0] * scale - off;
    g.y1 = roi[1] * scale - off;
    float x2 = roi[2] * scale - off;
    float y2 = roi[3] * scale - off;
    float rw = x2 - g.x1, rh = y2 - g.y1;
    if (!aligned) { if (rw < 1.0f) rw = 1.0f; if (rh < 1.0f) rh = 1.0f; }
    g.bin_h = rh / (float)PH;
    g.bin_w = rw / (float)PW;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

static float sample_y(const roi_geom_t* g, int ph, int iy) {
    return g->y1 + (float)ph * g->bin_h + ((float)iy + 0.5f) * g->bin_h / (float)g->grid_h;
}
static float sample_x(const roi_geom_t* g, int pw, int ix) {
    return g->x1 + (float)pw * g->bin_w + ((float)ix + 0.5f) * g->bin_w / (float)g->grid_w;
}

/* feat [NI,C,H,W]; rois [K,4] xyxy in input pixels; roi_img [K] image index; out [K,C,PH,PW]. */
int oracle_roi_align_fwd(const float* feat, const float* rois, const int32_t* roi_img, float* out,
                         int NI, int C, int H, int W, int K, int PH, int PW, float scale,
                         int sampling_ratio, int aligned) {
    (void)NI;
    for (int k = 0; k < K; ++k) {
        roi_geom_t g = geometry(rois + 4 * k, scale, PH, PW, sampling_ratio, aligned);
        const float* img = feat + (size_t)roi_img[k] * C * H * W;
        for (int c = 0; c < C; ++c) {
            const float* pl = img + (size_t)c * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < g.grid_h; ++iy)
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            neighbour_t n = locate(sample_y(&g, ph, iy), sample_x(&g, pw, ix), H, W);
                            if (n.y_low < 0) continue;
                            acc += n.w1 * pl[n.y_low * W + n.x_low] + n.w2 * pl[n.y_low * W + n.x_high] +
                                   n.w3 * pl[n.y_high * W + n.x_low] + n.w4 * pl[n.y_high * W + n.x_high];
                        }
                    out[(((size_t)k * C + c) * PH + ph) * PW + pw] = acc / g.count;
                }
        }
    }
    return 0;
}

/* dfeat must be zeroed by the caller. */
int oracle_roi_align_bwd(const float* dout, const float* rois, const int32_t* roi_img, float* dfeat,
                         int NI, int C, int H, int W, int K, int PH, int PW, float scale,
                         int sampling_ratio, int aligned) {
    (void)NI;
    for (int k = 0; k < K; ++k) {
        roi_geom_t g = geometry(rois + 4 * k, scale, PH, PW, sampling_ratio, aligned);
        float* img = dfeat + (size_t)roi_img[k] * C * H * W;
        for (int c = 0; c < C; ++c) {
            float* pl = img + (size_t)c * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float gr = dout[(((size_t)k * C + c) * PH + ph) * PW + pw] / g.count;
                    for (int iy = 0; iy < g.grid_h; ++iy)
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            neighbour_t n = locate(sample_y(&g, ph, iy), sample_x(&g, pw, ix), H, W);
                            if (n.y_low < 0) continue;
                            pl[n.y_low * W + n.x_low] += gr * n.w1;
                            pl[n.y_low * W + n.x_high] += gr * n.w2;
                            pl[n.y_high * W + n.x_low] += gr * n.w3;
                            pl[n.y_high * W + n.x_high] += gr * n.w4;
                        }
                }
        }
    }
    return 0;
}

/* Integer side of the contract (must be reproduced bit-exactly by the HIP kernel):
 * grid [K,2] = (grid_h, grid_w); nbr [K,PH,PW,4] = (y_low,x_low,y_high,x_high) of sample (iy,ix)=(0,0)
 * (all -1 when the grid is empty or the point is out of range). */
int oracle_roi_align_indices(const float* rois, int32_t* grid, int32_t* nbr, int H, int W, int K, int PH,
                             int PW, float scale, int sampling_ratio, int aligned) {
    for (int k = 0; k < K; ++k) {
        roi_geom_t g = geometry(rois + 4 * k, scale, PH, PW, sampling_ratio, aligned);
        grid[2 * k] = g.grid_h;
        grid[2 * k + 1] = g.grid_w;
        for (int ph = 0; ph < PH; ++ph)
            for (int pw = 0; pw < PW; ++pw) {
                int32_t* o = nbr + (((size_t)k * PH + ph) * PW + pw) * 4;
                if (g.grid_h <= 0 || g.grid_w <= 0) { o[0] = o[1] = o[2] = o[3] = -1; continue; }
                neighbour_t n = locate(sample_y(&g, ph, 0), sample_x(&g, pw, 0), H, W);
                o[0] = n.y_low; o[1] = n.x_low; o[2] = n.y_high; o[3] = n.x_high;
            }
    }
    return 0;
}
