/*
 * include/focus_amd.h -- C ABI of libfocus_amd.so (MI355X / gfx950 hot path of srv902/FOCUS).
 *
 * The reference has no FFI: its operator surface is the Python module API.
 * Each entry point below replaces the device work done by the cited reference call site; the
 * Python host mirror (focus_amd/slowfast/...) keeps the reference's signatures and calls these
 * through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); nothing here allocates,
 *     frees or synchronises; kernels are enqueued on `stream` (a hipStream_t, may be NULL);
 *   - tensors are dense row-major in the documented shape unless strides are passed;
 *   - `dtype` selects activation/weight storage: FOCUS_F32 or FOCUS_BF16 (accumulation, softmax and
 *     normalisation statistics are always fp32); LayerNorm affine parameters, biases and all
 *     parameter gradients are fp32 in both modes;
 *   - return value: 0 on success, negative focus_status on error (focus_strerror() names it).
 */
#ifndef FOCUS_AMD_H
#define FOCUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum focus_dtype { FOCUS_F32 = 0, FOCUS_BF16 = 1, FOCUS_FP8_E4M3 = 2 /* OCP e4m3fn codes: weights (B operand) only */ };

enum focus_status {
    FOCUS_OK = 0,
    FOCUS_ERR_SHAPE = -1,      /* inconsistent or unsupported shape            */
    FOCUS_ERR_DTYPE = -2,      /* unsupported dtype combination                */
    FOCUS_ERR_ALIGN = -3,      /* pointer / stride alignment not met           */
    FOCUS_ERR_LAUNCH = -4,     /* hipGetLastError() reported a launch failure  */
    FOCUS_ERR_NULL = -5,       /* required pointer is NULL                     */
    FOCUS_ERR_WORKSPACE = -6   /* workspace too small                          */
};

const char* focus_strerror(int status);
/* ABI version of this library (bumped on any signature change). */
int focus_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C = epi(alpha * A.B [+ bias]) [+ residual]      (fp32 accumulate)
 * Replaces every nn.Linear / bmm / einsum contraction on the path (attention.py:506,524-529,536-537,
 * 555; common.py:26-34; orvit.py:59-72; steve.py:61-62,75-76,83; stem_helper.py:318 as im2col GEMM)
 * and their autograd backward.  A is [M,K], B is [K,N], C is [M,N]; all three are addressed by
 * element strides (rs = row stride, cs = column stride) and a two-level batch (batch0 x batch1).
 * The MFMA path is taken when dtype is bf16 and both A and B are contiguous along K (csA==1,
 * rsB==1), 16-byte aligned; everything else runs the generic tiled kernel.
 * ----------------------------------------------------------------------------------------------*/
enum focus_epilogue {
    FOCUS_EPI_NONE = 0,
    FOCUS_EPI_GELU = 1,    /* C = gelu_erf(v); if aux != NULL the pre-activation v is stored in aux */
    FOCUS_EPI_RELU = 2,
    FOCUS_EPI_TANH = 3,
    FOCUS_EPI_DGELU = 4,   /* C = v * gelu'(aux)      aux = saved pre-activation                    */
    FOCUS_EPI_DRELU = 5,   /* C = v * (aux > 0)       aux = saved output                            */
    FOCUS_EPI_DTANH = 6    /* C = v * (1 - aux^2)     aux = saved output                            */
};

typedef struct focus_gemm_desc {
    int32_t M, N, K;
    int32_t batch0, batch1;
    const void* A; int64_t rsA, csA, bsA0, bsA1;
    const void* B; int64_t rsB, csB, bsB0, bsB1;
    void* C;       int64_t rsC, csC, bsC0, bsC1;
    const float* bias;        /* [N] fp32 or NULL (applied before the activation)                   */
    const void* residual;     /* same dtype/strides as C, added after the activation, or NULL       */
    void* aux;                /* same dtype/strides as C; see focus_epilogue                        */
    float alpha;
    int32_t accumulate;       /* 1: C += result (C must be fp32)                                    */
    int32_t epilogue;         /* enum focus_epilogue                                                */
    int32_t dtype_ab;         /* storage of A and B                                                 */
    int32_t dtype_c;          /* storage of C, residual and aux                                     */
    int32_t dtype_b;          /* 0: B is stored like A.  FOCUS_FP8_E4M3: B holds e4m3 codes (1 byte per element,
                                 strides in elements), A is bf16: the "fp8 weights, bf16 activations" GEMM     */
    int32_t pad_;
    const float* b_scale;     /* DEVICE scalar: the per-tensor scale of an fp8 B (C = epi(alpha * b_scale * A.B ...)) */
} focus_gemm_desc;

int focus_gemm(const focus_gemm_desc* desc, void* stream);

/* Kernel family the last focus_gemm() call of this thread dispatched to (measurement attribution only). */
enum focus_gemm_kernel {
    FOCUS_GEMM_KERNEL_GENERIC = 0,   /* gemm_generic.hip (fp32 / odd strides)                       */
    FOCUS_GEMM_KERNEL_NT = 1,        /* gemm_mfma.hip uniform 128x128 (few tiles, split-K)          */
    FOCUS_GEMM_KERNEL_NT_WS = 2,     /* gemm_mfma_ws.hip wave-specialised (the step's dominant kernel) */
    FOCUS_GEMM_KERNEL_TN = 3,        /* gemm_mfma_tn*.hip (weight gradients)                        */
    FOCUS_GEMM_KERNEL_NT_SMALL = 4   /* gemm_mfma_small.hip (M <= 1024 rows: recurrent / motion-stream Linears) */
};
int focus_gemm_last_kernel(void);

/* Tuning hook: force the row-tile height of the wave-specialised NT kernel (128 or 192; 0 = automatic choice by the
 * modelled rounds-x-tile-time cost).  Used by tools/gemm_tile_ab.py for A/B timing inside one process. */
int focus_gemm_tile_override(int bm);

/* Weight-gradient form (A strided along the reduction: rsA == 1, csB == 1, bf16 in, fp32 out): the long reduction is
 * split over workgroups.  With desc->aux == NULL the partial sums are added to a zero-initialised C with fp32
 * atomics (desc->accumulate must be 1); with desc->aux pointing to focus_gemm_tn_workspace_bytes(M, N, K) bytes the
 * partials are stored to per-split slabs and summed by a second kernel (no atomics, bitwise reproducible, C is
 * overwritten). */
size_t focus_gemm_tn_workspace_bytes(int M, int N, int K);
/* Batched form (desc->batch0 == 1, batch1 = batch independent products whose outputs are stacked densely:
 * rsC == N, bsC1 == M*N; operands offset by bsA1 / bsB1 per batch): slab mode only, aux of this many bytes. */
size_t focus_gemm_tn_batched_workspace_bytes(int M, int N, int K, int batch);

/* nn.Linear backward w.r.t. its parameters in one pass over dy (replaces autograd's dy^T @ x and dy.sum(0) of
 * common.py:26-34 / attention.py:506,536-537,555):  dw[N,K] = dy[M,N]^T . x[M,K],  db[N] = sum_m dy[m,:]  (db may be
 * NULL).  bf16 operands, fp32 results, both overwritten.  The column sums ride on the matrix pipe of the
 * weight-gradient kernel (ones^T . dy) when the wave-specialised kernel takes the shape; otherwise focus_colsum runs.
 * ws: focus_linear_wgrad_workspace_bytes(N, K, M) bytes.  N % 8 == 0, K % 8 == 0, 16-byte aligned rows. */
size_t focus_linear_wgrad_workspace_bytes(int N, int K, int M);
int focus_linear_wgrad(const void* dy, const void* x, float* dw, float* db, void* ws, size_t ws_bytes, int M, int N,
                       int K, int64_t ld_dy, int64_t ld_x, int dtype, void* stream);

/* y[M,N] = act(x[M,K] . w[N,K]^T + bias) + residual -- nn.Linear forward (thin wrapper over focus_gemm). */
int focus_linear_fwd(const void* x, const void* w, const float* bias, const void* residual, void* y,
                     void* aux, int M, int N, int K, int epilogue, int dtype, void* stream);

/* out[n] (+)= sum_m x[m,n]   (bias gradients)  x is `dtype`, out fp32.  */
int focus_colsum(const void* x, float* out, int M, int N, int64_t row_stride, int accumulate, int dtype,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last axis (video_model_builder.py:1129 eps 1e-6; steve.py:35-37 eps 1e-5).
 * ----------------------------------------------------------------------------------------------*/
int focus_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int rows, int D, float eps, int dtype, void* stream);
/* dgamma/dbeta are written (not accumulated); `partial` is a [2, nblk, D] fp32 scratch with
 * nblk = focus_layernorm_bwd_blocks(rows).  dres (may be NULL): gradient arriving on the residual path around a
 * pre-norm block (x -> x + f(LN(x)), attention.py:116-126); it is added into dx in the same pass, replacing the
 * separate accumulation autograd would do.  dgamma == dbeta == NULL: only `partial` is written -- partial[0] holds nblk row
 * vectors whose sum is dgamma, partial[1] likewise dbeta -- for a caller that applies the same LayerNorm many times and
 * reduces all applications' partials in one pass at the end (STEVE's slot loop: 213 applications per step). */
int focus_layernorm_bwd_blocks(int rows);
int focus_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* partial,
                        int rows, int D, int dtype, void* stream);
/* The same two kernels over ROW BLOCKS: x (and, backward, dx) is a set of blocks of rows_per_block dense rows whose
 * starts are block_stride elements apart -- frame t of a [B,T,N,D] video tensor read and differentiated in place (pointer
 * to frame t, rows_per_block = N, block_stride = T*N*D, rows = B*N): no per-frame copies either way (steve.py:60, :68).
 * y, dy, mean, rstd are dense [rows, D]. */
int focus_layernorm_fwd_blocks(const void* x, int rows_per_block, int64_t block_stride, const float* gamma,
                               const float* beta, void* y, float* mean, float* rstd, int rows, int D, float eps, int dtype,
                               void* stream);
int focus_layernorm_bwd_blocks_strided(const void* dy, const void* x, int rows_per_block, int64_t block_stride,
                                       const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                       float* dbeta, float* partial, int rows, int D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row softmax over segments:  x viewed as [rows, L] with row stride `stride`; in place allowed.
 * bwd: dx = y * (dy - sum(dy*y)).   Used by the fp32 (unfused) attention path, the motion-stream
 * joint attention (attention.py:380-381), the cls row (attention.py:436-440) and the slot predictor
 * (transformer.py:44).
 * ----------------------------------------------------------------------------------------------*/
int focus_softmax_fwd(const void* x, void* y, int64_t rows, int L, int64_t stride, float scale, int dtype,
                      void* stream);
int focus_softmax_bwd(const void* dy, const void* y, void* dx, int64_t rows, int L, int64_t stride,
                      float scale, int dtype, void* stream);
/* Causal variant (STEVE/transformer.py:145-166, self_attn_mask = triu(1)): row r sees columns 0 .. r % period, the
 * masked tail of y is exact zeros; its backward is focus_softmax_bwd. */
int focus_softmax_causal_fwd(const void* x, void* y, int64_t rows, int L, int64_t stride, int period, float scale,
                             int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Trajectory attention (attention.py:499-557).  qkv is the output of the fused qkv Linear,
 * [B, N, 3C] with N = 1 + F*P, channel order (q|k|v) x (head, d).
 *   space step (:514-535): cls_out [B,C]; xt = x~ [B,S,F,C]; xdiag [B,S,C]; lse [B,h,S,F] and
 *   cls_lse [B,h] (fp32, saved for backward).
 *   time step (:538-549, use_original_code=True): q2 [B,S,C] (un-scaled proj_q output),
 *   k2 [B,S,F,C], xt -> out [B,S,C]; attn2 [B,h,S,F] fp32 saved for backward.
 * Workspace sizes are returned by the *_workspace_bytes queries (0 when none is needed).
 * ----------------------------------------------------------------------------------------------*/
size_t focus_traj_space_workspace_bytes(int B, int F, int P, int heads, int d, int dtype, int backward);
int focus_traj_space_fwd(const void* qkv, void* xt, void* xdiag, void* cls_out, float* lse, float* cls_lse,
                         void* workspace, size_t workspace_bytes, int B, int F, int P, int heads, int d,
                         int dtype, void* stream);
/* dqkv [B,N,3C] is fully written. dxt / dxdiag / dcls are the cotangents of xt / xdiag / cls_out. */
int focus_traj_space_bwd(const void* qkv, const void* xt, const void* cls_out, const float* lse,
                         const float* cls_lse, const void* dxt, const void* dxdiag, const void* dcls,
                         void* dqkv, void* workspace, size_t workspace_bytes, int B, int F, int P,
                         int heads, int d, int dtype, void* stream);
/* out_bstride / dout_bstride: elements between consecutive batches of out / dout (S*C when dense; (S+1)*C when the
 * rows live in tokens 1.. of a [B,1+S,C] buffer whose token 0 is the cls row -- saves the concatenation of
 * attention.py:551 and the slice copy of its backward). */
int focus_traj_time_fwd(const void* q2, const void* k2, const void* xt, void* out, int64_t out_bstride, float* attn2,
                        int B, int S, int F, int heads, int d, int dtype, void* stream);
int focus_traj_time_bwd(const void* q2, const void* k2, const void* xt, const float* attn2, const void* dout,
                        int64_t dout_bstride, void* dq2, void* dk2, void* dxt, int dxt_accumulate, int B, int S, int F,
                        int heads, int d, int dtype, void* stream);

/* Temporal step WITHOUT k2 = proj_kv(x~)[:, :C] in HBM (attention.py:536-549, use_original_code=True).  The logits are
 *   scale * q2[s,h,:] . (Wk[h] x~[s,f,:] + bk[h])  =  scale * (Wk[h]^T q2[s,h,:]) . x~[s,f,:]  + (a term constant in f),
 * and the softmax over f is shift invariant, so k2 [B,S,F,C] (the block's largest GEMM: 8x the tokens) and dk2 are
 * never formed; u[s,h,:] = Wk[h]^T q2[s,h,:] is produced per 64-channel chunk on chip (csrc/traj_time2.hip).
 * bf16, head dim 64, heads <= 16, F in {4, 8, 16}.  wkT: [C rows (input channel c)][ldw] bf16 whose columns 0..C-1 hold
 * Wk^T (the transposed shadow of proj_kv.weight).  q2 [B,S,C] is the UN-scaled proj_q output.
 *   fwd: out rows [B,S,C] at batch stride out_bstride (see focus_traj_time_fwd); attn2 [B,S,h,F] fp32 (saved);
 *        ws: focus_traj_time2_workspace_bytes() bytes of scratch.
 *   bwd: dxt [B,S,F,C] (fully written: a*dout + sum_h dl*u); g [h,B*S,C] bf16 = d(loss)/d(u), head-major so that each
 *        head's slice is a dense matrix, from which the caller forms  dq2[:, h*64+dd] = sum_c g[h,:,c] Wk[h*64+dd, c]
 *        and  dWk[h*64+dd, c] = sum_s q2[s,h*64+dd] g[h,s,c]  (two batched GEMMs over the heads); dl [B,S,F,16] bf16 scratch.  proj_kv.bias gets its exact zero gradient. */
size_t focus_traj_time2_workspace_bytes(int B, int S, int F, int heads, int d);
int focus_traj_time2_fwd(const void* q2, const void* xt, const void* wkT, int64_t ldw, void* out, int64_t out_bstride,
                         float* attn2, void* ws, size_t ws_bytes, int B, int S, int F, int heads, int d, int dtype,
                         void* stream);
int focus_traj_time2_bwd(const void* q2, const void* xt, const void* wkT, int64_t ldw, const float* attn2,
                         const void* dout, int64_t dout_bstride, void* dxt, void* g, void* dl, int B, int S, int F,
                         int heads, int d, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RoIAlign over patch-token feature maps (ORViT/utils.py:58-75 -> torchvision.ops.roi_align with
 * output_size=(H,W), sampling_ratio=-1, aligned=True).  Channels-last on both sides:
 *   feat  image i = (b, t) = (i / imgs_per_batch, i % imgs_per_batch) starts at feat + b*batch_stride + t*img_stride
 *         (elements): the tokens are read in place from the residual stream [B,1+T*H*W,C] (pointer to token 1,
 *         img_stride = H*W*C, imgs_per_batch = T, batch_stride = (1+T*H*W)*C); a dense [NI,H*W,C] map is
 *         imgs_per_batch = NI, img_stride = H*W*C;
 *   rois  [K,4] fp32 xyxy in input pixels, roi_img [K] int32 image index;
 *   out   [K, PH*PW, C].
 * The integer side (sampling grid size, neighbour indices) is bit-exact with oracle/roi_align_ref.c.
 * ----------------------------------------------------------------------------------------------*/
int focus_roi_align_fwd(const void* feat, int64_t img_stride, int imgs_per_batch, int64_t batch_stride,
                        const float* rois, const int32_t* roi_img, void* out, int NI, int C, int H, int W, int K,
                        int PH, int PW, float spatial_scale, int sampling_ratio, int aligned, int relu, int dtype,
                        void* stream);
/* dfeat: image maps addressed like feat, `dtype`, every map fully written (no zero-initialisation needed; rows between
 * the maps -- the cls row of a token buffer -- are not touched).  Maps up to 16x16 with up to
 * 14x14 bins use the separable form dfeat = sum_rois Ay . dout . Ax^T (no atomics, accumulators in registers); other
 * shapes go through fp32 atomics into `ws` (focus_roi_align_bwd_workspace_bytes, 0 for the separable path) + a cast
 * and need the dense addressing (FOCUS_ERR_SHAPE otherwise).
 * relu != 0 (forward): out = max(RoIAlign, 0) -- the ReLU that follows when patch_to_d's first Linear is applied BEFORE
 * sampling (it has no bias, so it commutes with the bilinear sampling: SURVEY a10); relu_out (backward, may be NULL,
 * separable path only): that saved output, used as the ReLU mask on dout. */
size_t focus_roi_align_bwd_workspace_bytes(int NI, int C, int H, int W, int PH, int PW);
int focus_roi_align_bwd(const void* dout, const void* relu_out, const float* rois, const int32_t* roi_img, void* dfeat, int64_t img_stride,
                        int imgs_per_batch, int64_t batch_stride, void* ws, size_t ws_bytes, int NI, int C, int H, int W, int K, int PH, int PW, float spatial_scale,
                        int sampling_ratio, int aligned, int dtype, void* stream);
/* Debug/parity export of the integer side: grid [K,2], nbr [K,PH,PW,4] (int32). */
int focus_roi_align_indices(const float* rois, int32_t* grid, int32_t* nbr, int H, int W, int K, int PH,
                            int PW, float spatial_scale, int sampling_ratio, int aligned, void* stream);

/* max over the cells of each RoI (orvit.py:138): x [K, cells, C] -> y [K, C], arg [K, C] int32. */
int focus_cell_amax_fwd(const void* x, void* y, int32_t* arg, int K, int cells, int C, int dtype,
                        void* stream);
/* dx [K, cells, C] is fully written (zeros except the arg-max cell). */
int focus_cell_amax_bwd(const void* dy, const int32_t* arg, void* dx, int K, int cells, int C, int dtype,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * ORViT token plumbing (ORViT/orvit.py:145-147, :152-157, :165-169), one pass each way instead of cat / slice / add chains.
 * x   [B, 1+T*HW, C]     residual stream (cls row + patch tokens)
 * obj [B, T, O, C]       object tokens
 * all [B, 1+T*(HW+O), C] cls row, then per frame its HW patch tokens followed by its O object tokens
 * assemble: all = cat(cls, cat(patch.view(B,T,HW,C), obj, dim=2).flatten(1,2)); the adjoint scatters d(all) into
 *           dx (every row written) and dobj.  Pure row copies: any dtype, C * elem_size % 16 == 0.
 * merge:    out[b,0] = x[b,0] + s_b*y[b,0];  out[b,1+t*HW+p] = x[..] + s_b*(y[b,1+t*(HW+O)+p] + mm[b,t*HW+p])
 *           (y = attention output over `all`, mm = motion-stream MLP output or NULL, s = per-sample stochastic-depth
 *           scale or NULL for 1); adjoint: dy (object rows zero), dmm = s*dout[:,1:] (or NULL); dx is dout itself.
 * ----------------------------------------------------------------------------------------------*/
int focus_orvit_assemble(const void* x, const void* obj, void* all, int B, int T, int HW, int O, int C, int dtype,
                         void* stream);
int focus_orvit_assemble_bwd(const void* dall, void* dx, void* dobj, int B, int T, int HW, int O, int C, int dtype,
                             void* stream);
int focus_orvit_merge(const void* x, const void* y, const void* mm, const float* scale, void* out, int B, int T, int HW,
                      int O, int C, int dtype, void* stream);
int focus_orvit_merge_bwd(const void* dout, const float* scale, void* dy, void* dmm, int B, int T, int HW, int O, int C,
                          int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Box layout (ORViT/utils.py:8-28 -> layout.py:28-63,98-130,205-237), closed form of SURVEY.md A4:
 *   out[n,y,x,:] = sum_o keep_o * vecs[n,o,:] * w((lin_y - y0_o)/y1_o) * w((lin_x - x0_o)/x1_o)
 * boxes [NF,O,4] fp32 cxcywh; vecs [NF,O,C]; out [NF,H*W,C].   NF = B*T frames.
 * ----------------------------------------------------------------------------------------------*/
int focus_box_layout_fwd(const void* vecs, const float* boxes, void* out, int NF, int O, int C, int H,
                         int W, int dtype, void* stream);
int focus_box_layout_bwd(const void* dout, const float* boxes, void* dvecs, int NF, int O, int C, int H,
                         int W, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Slot-attention inverted softmax + weighted mean (steve.py:76-83) for one frame t, one iteration:
 *   logits[b,n,k] = k_t[b,n,:].q[b,k,:]; attn_vis = softmax_k; a = (attn_vis+eps)/sum_n; upd = a^T v_t
 * k_t, v_t [B,N,D] with batch stride `kv_bstride` elements (frame t of [B,T,N,D] read in place);
 * q [B,K,D]; attn_vis [B,N,K] (stride attn_bstride); upd [B,K,D]; colsum [B,K] fp32 (saved).
 * K <= 32, D <= 256.  partial: fp32 scratch of focus_slot_attn_workspace_bytes().
 * ----------------------------------------------------------------------------------------------*/
size_t focus_slot_attn_workspace_bytes(int B, int N, int K, int D);
int focus_slot_attn_fwd(const void* k_t, const void* v_t, int64_t kv_bstride, const void* q, void* attn_vis,
                        int64_t attn_bstride, void* upd, float* colsum, void* partial, size_t partial_bytes,
                        int B, int N, int K, int D, float eps, int dtype, void* stream);
/* dattn_vis may be NULL (no cotangent on the visualised attention).  dk_t/dv_t written at the same
 * strides as k_t/v_t (accumulate=1: added to the existing contents).
 * wl != NULL (focus_slot_kv_grad_ok): d(k_t), d(v_t) are NOT formed here (dk_t, dv_t may be NULL); the launch writes its
 * rows of w = (attn + eps) / colsum and d(logits) to wl [B,N,32] bf16 (slots 0..15 | 16..31) for focus_slot_kv_grad. */
int focus_slot_attn_bwd(const void* k_t, const void* v_t, int64_t kv_bstride, const void* q,
                        const void* attn_vis, int64_t attn_bstride, const float* colsum, const void* upd,
                        const void* dupd,
                        const void* dattn_vis, void* dk_t, void* dv_t, int accumulate, void* dq, void* partial,
                        size_t partial_bytes, int B, int N, int K, int D, float eps, int dtype, void* wl, void* stream);
/* d(k_t), d(v_t) of one frame for all `iters` (<= 4) corrector iterations that read it (steve.py:68-83), from the wl rows
 * of their backward launches and their q / dupd [B,K,D]:  dk = sum_i dlogits_i . q_i,  dv = sum_i w_i . dupd_i  (two MFMA
 * products with the iterations stacked along the reduction).  bf16, K <= 16, D in {64,128,192,256}.  dk_t / dv_t rows are
 * kv_ld elements apart (>= D): with dv_t = dk_t + D and kv_ld = 2 D the two gradients form one [B,N,2D] matrix [dk | dv],
 * which lets the projections' backward run as one product each (d(input) = [dk|dv].[Wk;Wv], d[Wk;Wv] = [dk|dv]^T.x). */
int focus_slot_kv_grad_ok(int K, int D, int dtype, int iters);
int focus_slot_kv_grad(const void* wl0, const void* wl1, const void* wl2, const void* wl3, const void* q0, const void* q1,
                       const void* q2, const void* q3, const void* du0, const void* du1, const void* du2, const void* du3,
                       int iters, void* dk_t, void* dv_t, int64_t kv_bstride, int64_t kv_ld, int B, int N, int K, int D,
                       int dtype, void* stream);

/* Multi-head attention over a handful of tokens (STEVE's slot predictor, transformer.py:4-49: K = 11 slots, 4 heads of 48),
 * one launch each way: q [B,N,heads*d], k, v [B,M,heads*d] with rows ldq / ldk / ldv elements apart and no gap between
 * clips (ld = heads*d: dense; the three may be the column blocks of one [B*N, 3*heads*d] projection output); att
 * [B,heads,N,M] = softmax_rows(scale q k^T) is stored for the backward; out, dout [B,N,heads*d] dense = att v.  dq, dk, dv
 * are written at the strides of q, k, v.  N, M <= 16, d <= 64 (focus_small_attn_ok); no mask, no dropout. */
int focus_small_attn_ok(int N, int M, int d);
int focus_small_attn_fwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, void* att,
                         void* out, int B, int heads, int N, int M, int d, float scale, int dtype, void* stream);
int focus_small_attn_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, const void* att,
                         const void* dout, void* dq, void* dk, void* dv, int B, int heads, int N, int M, int d, float scale,
                         int dtype, void* stream);

/* nn.GRUCell gate math (STEVE/utils.py:107-118): gi, gh [R,3D] gate pre-activations, h [R,D] -> hn.
 * b_ih, b_hh (fp32 [3D]; both or neither): with them gi / gh arrive WITHOUT bias (the cell's two Linear products run as one
 * batched bias-free launch), the kernel adds the biases and writes the biased values back into gi / gh -- what the
 * backward reads.  NULL: gi / gh already carry their biases and are only read. */
int focus_gru_gates_fwd(void* gi, void* gh, const void* h, void* hn, const float* b_ih, const float* b_hh, int R, int D,
                        int dtype, void* stream);
/* dh = the direct part dhn * z.  zero_out (may be NULL): an [R,D] buffer cleared by the same launch -- with dh it forms the
 * [2,R,D] residual operand of the batched product [d(input) | d(h)] = [dgi | dgh] . [W_ih | W_hh] + [0 | dh]. */
int focus_gru_gates_bwd(const void* gi, const void* gh, const void* h, const void* dhn, void* dgi, void* dgh,
                        void* dh, void* zero_out, int R, int D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Patch embedding as GEMM (stem_helper.py:317-320): im2col of x [B,Cin,T,H,W] fp32 (loader output)
 * for a kernel == stride patch (kt,kh,kw) -> cols [B*T'*H'*W', Cin*kt*kh*kw] in `dtype`,
 * rows in (t,h,w) token order, columns in Conv3d weight order (c,dt,dh,dw).
 * ----------------------------------------------------------------------------------------------*/
int focus_im2col_patches(const float* x, void* cols, int B, int Cin, int T, int H, int W, int kt, int kh,
                         int kw, int dtype, void* stream);

/* tokens[b,0,:] = cls + pos[0]; tokens[b,1+t*P+p,:] = patch[b,t*P+p,:] + pos[1+p] + temp[t]
 * (video_model_builder.py:1281-1320). cls [C], pos [1+P,C], temp [T,C] fp32. */
int focus_embed_assemble(const void* patch, const float* cls, const float* pos, const float* temp,
                         void* tokens, int B, int T, int P, int C, int dtype, void* stream);

/* Label-smoothing cross entropy (losses.py:53-59): loss_rows [R] fp32, dlogits [R,Ncls] fp32
 * (= d mean-loss / d logits).  logits fp32. */
int focus_xent_ls(const float* logits, const int64_t* target, float* loss_rows, float* dlogits, int R,
                  int Ncls, float smoothing, void* stream);

/* dst[c, r] = src[r, c] for src [R, Cc] with row stride src_ld; dst has row stride dst_ld >= R and
 * columns R..dst_ld-1 are zero-filled (so a transposed operand can feed the K-contiguous MFMA GEMM).
 * src may be fp32 or bf16 (src_dtype); dst is `dst_dtype`. Batched over `batch` with the given strides. */
int focus_transpose_pad(const void* src, int src_dtype, int64_t src_ld, int64_t src_bstride, void* dst,
                        int dst_dtype, int64_t dst_ld, int64_t dst_bstride, int R, int Cc, int batch,
                        void* stream);

/* bf16 working copies of fp32 master weights, all tensors of a model in ONE launch (what autocast does per use in
 * the reference, train_net.py:84 `torch.cuda.amp.autocast`): for every item, dst[r,c] = bf16(src[r,c]) (row-major,
 * optional) and dstT[c,r] = bf16(src[r,c]) (the K-contiguous operand of the dX GEMM, optional).
 * `items` is a DEVICE array of n_items focus_shadow_item; rows % 4 == 0 and cols % 4 == 0, 16-byte aligned src. */
typedef struct focus_shadow_item {
    const float* src; void* dst; void* dstT;
    int32_t rows, cols;
} focus_shadow_item;
int focus_shadow_refresh(const focus_shadow_item* items, int n_items, int max_rows, int max_cols, void* stream);

/* The optimizer step of train_net.py:108-120 for ALL parameters of a model in two launches (optim.hip): global gradient
 * norm (clip_grad_norm_, train_net.py:112-117; max_norm <= 0: no clipping), AdamW update (torch.optim.AdamW as built by
 * optimizer.py:137-145: decoupled weight decay, bias correction, no amsgrad) of the fp32 masters and moments, and the bf16
 * working copies of the updated weights (dst row-major, dstT transposed; either may be NULL).
 * `items`: DEVICE array; `grads`: DEVICE array of the n_items fp32 gradient pointers (kept apart from `items`: they are
 * the only pointers that change from step to step); unit0 = first work unit of the item (prefix sum of focus_adamw_units over the items);
 * tile_mode 1 needs rows % 4 == 0, cols % 4 == 0 and 16-byte aligned tensors (dstT only in tile mode).
 * `groups`: DEVICE [n_groups][2] = (lr, weight_decay).  `steps`: DEVICE [n_items] fp32 step counters, incremented here.
 * `total_norm`: DEVICE scalar receiving the gradient norm before clipping, or NULL.
 * write_clipped_grads: g *= clip coefficient is stored back (what clip_grad_norm_ leaves in .grad). */
typedef struct focus_adamw_item {
    float* p; float* m; float* v; void* dst; void* dstT;
    int32_t rows, cols, group, unit0, tile_mode, pad_;
} focus_adamw_item;
int focus_adamw_units(int rows, int cols, int tile_mode);
size_t focus_adamw_workspace_bytes(void);
int focus_adamw_step(const focus_adamw_item* items, float* const* grads, int n_items, int n_units, const float* groups, float* steps,
                     void* workspace, size_t workspace_bytes, float* total_norm, double beta1, double beta2, float eps,
                     float max_norm, int write_clipped_grads, void* stream);

/* OCP FP8 E4M3 working copies of fp32 master weights with one scale per tensor (BASELINE configs[4]; the reference's
 * EK_ORVIT_MF_HR.yaml trains fp16-autocast: these replace autocast's per-use fp16 weight casts, train_net.py:84):
 * scale = amax|w| / 448, dst[r,c] = e4m3(w[r,c] / scale) (row-major, optional), dstT[c,r] = the same code transposed
 * (the K-contiguous B operand of the dX GEMM, optional), *scale = the fp32 scale (1 for an all-zero tensor).
 * `items`: DEVICE array; rows % 4 == 0, cols % 4 == 0, 16-byte aligned src.  amax_scratch: n_items x 4 bytes. */
typedef struct focus_fp8_item {
    const float* src; void* dst; void* dstT; float* scale;
    int32_t rows, cols;
} focus_fp8_item;
int focus_fp8_refresh(const focus_fp8_item* items, int n_items, int max_rows, int max_cols, void* amax_scratch, void* stream);

/* Weight and bias gradients of up to 8 nn.Linear layers in ONE launch (gemm_tn_group.hip; autograd of attention.py:506,
 * 536,555 and common.py:26-34 inside one block): dw_p[N,K] = dy_p[M,N]^T . x_p[M,K] (fp32, dense), db_p[N] += column sums
 * of dy_p (fp32 atomics: the caller zeroes db; NULL = no bias).  bf16 operands, row strides ld_dy / ld_x in elements;
 * N, K, strides multiples of 8; 16-byte aligned pointers.  `items` is a HOST array. */
typedef struct focus_wgrad_item {
    const void* dy; const void* x; float* dw; float* db;
    int64_t ld_dy, ld_x;
    int32_t M, N, K, pad_;
} focus_wgrad_item;
int focus_linear_wgrad_group_units(const focus_wgrad_item* items, int n_items);
/* The launch's unit list (n_units x 8 bytes, HOST memory): a function of the (N, K) sequence only; the caller keeps a DEVICE
 * copy per shape signature and hands it to focus_linear_wgrad_group. */
int focus_linear_wgrad_group_plan(const focus_wgrad_item* items, int n_items, void* host_units, size_t bytes);
int focus_linear_wgrad_group(const focus_wgrad_item* items, int n_items, const void* dev_units, void* stream);

/* The per-iteration tail of the STEVE slot update in one launch (slot_tail.hip; steve.py:72-75,85-93, STEVE/utils.py:107-118):
 * [do_gru] GRU cell on (upd, h) -> hn;  [do_mlp] s = hn + W2 relu(W1 LN1(hn) + b1) + b2;  [do_q] q = Wq LN2(slots) where slots
 * is s, hn or (do_gru == 0: "q only") the input h itself.  R rows of D (= 192) channels, hidden width H (= 768); bf16
 * activations and weights ([out, in] row-major, the bf16 working copies), fp32 biases / LayerNorm parameters / statistics.
 * Every intermediate a backward needs is an output: g [2,R,3D] (gate pre-activations incl. bias), hn, y = LN1(hn) with
 * mean1 / rstd1 [R], a = relu(..) [R,H], s, sn = LN2(slots) with mean2 / rstd2, q. */
typedef struct focus_slot_tail_args {
    int32_t R, D, H, do_gru, do_mlp, do_q;
    float ln1_eps, ln2_eps;
    const void* upd; const void* h;
    const void* w_ih; const void* w_hh; const float* b_ih; const float* b_hh;
    const float* ln1_g; const float* ln1_b; const void* w1; const float* b1; const void* w2; const float* b2;
    const float* ln2_g; const float* ln2_b; const void* wq;
    void* g; void* hn; void* y; float* mean1; float* rstd1; void* a; void* s;
    void* sn; float* mean2; float* rstd2; void* q;
} focus_slot_tail_args;
int focus_slot_tail_ok(int D, int H, int dtype);
int focus_slot_tail_fwd(const focus_slot_tail_args* args, void* stream);
/* The same chain backwards in one launch (autograd of steve.py:84-93 / utils.py:107-118 per iteration), against the TRANSPOSED
 * bf16 weight copies ([in][out]).  Inputs: dout (gradient of the slots; may be NULL = 0), dq (with do_q), the forward's saved
 * tensors (cur = the slots LayerNorm_slots normalised: s, or hn without the MLP, or h without the GRU).  Outputs: dupd, dh
 * [R,D]; the dY rows the weight gradients are later formed from: ds [R,D] (fc2; also with do_q alone), dz [R,H] (fc1), dg
 * [2,R,3D] (W_ih, W_hh) (dq itself is Wq's); LayerNorm parameter partials part1 / part2 [2][focus_slot_tail_bwd_blocks(R)][D]
 * fp32 ([0] = d gamma, [1] = d beta; sum over the blocks). */
typedef struct focus_slot_tail_bwd_args {
    int32_t R, D, H, do_gru, do_mlp, do_q;
    const void* dout; const void* dq;
    const void* h; const void* g; const void* hn; const void* a; const void* cur;
    const float* mean1; const float* rstd1; const float* mean2; const float* rstd2;
    const float* ln1_g; const float* ln2_g;
    const void* w_ih_t; const void* w_hh_t; const void* w1_t; const void* w2_t; const void* wq_t;
    void* dupd; void* dh; void* ds; void* dz; void* dg; float* part1; float* part2;
    void* ws_dsn; void* ws_dy1; void* ws_res;     /* [R, D] bf16 scratch between the launches of the staged form (do_q / do_mlp / do_gru) */
} focus_slot_tail_bwd_args;
int focus_slot_tail_bwd_blocks(int R);
int focus_slot_tail_bwd(const focus_slot_tail_bwd_args* args, void* stream);

/* Multi-head attention without the [Nq, Nk] probabilities in memory (flash_attn.hip) -- the STEVE decoder's causal
 * self-attention over the image tokens of a frame: STEVE/transformer.py:23-49 (MultiHeadAttention.forward: scale, mask,
 * softmax, dropout on the probabilities, product with v) with the upper-triangular mask of :131-132 / :149-151, and its autograd.
 *   out[b, n, h*d:(h+1)*d] = dropout(softmax_k(scale * q.k + causal mask)) v        lse[b, h, n] = log sum_k exp(scale * q.k)
 * q / k / v / out / dout / dq / dk / dv: bf16 rows of `heads * d` channels, row strides ld*, batch strides bs* (elements; the
 * three inputs may be column blocks of one projection output).  d in {32, 48, 64}; causal needs Nq == Nk.
 * drop_thr = round(p * 65536) (0: no dropout): element (b*heads + h, query, key) is kept iff the 16-bit half (key & 1) of
 * lowbias32(seed[0] ^ (bh * 0x9E3779B1) ^ (query * 0x85EBCA77) ^ ((key >> 1) * 0xC2B2AE3D)) is >= drop_thr, kept values are
 * scaled by 65536 / (65536 - drop_thr); `seed` is read on the device, so a draw can be captured in a graph.
 * bwd: delta [B, heads, Nq] fp32 is scratch (written by the dQ kernel, read by the dK/dV kernel); dq, dk, dv fully written. */
typedef struct focus_flash_args {
    const void* q; const void* k; const void* v; void* out; float* lse;
    const void* dout; float* delta; void* dq; void* dk; void* dv;
    const uint32_t* seed;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int64_t bsq, bsk, bsv, bso, bsdo, bsdq, bsdk, bsdv;
    int32_t B, heads, Nq, Nk, d, dtype, causal;
    uint32_t drop_thr;
    float scale;
    int32_t pad_;
} focus_flash_args;
int focus_flash_attn_ok(int Nq, int Nk, int d, int dtype, int causal);
int focus_flash_attn_fwd(const focus_flash_args* args, void* stream);
int focus_flash_attn_bwd(const focus_flash_args* args, void* stream);

/* Both consumers of g = d(loss)/d(u) of the re-associated temporal step in one pass over g (traj_time2_gw.hip; autograd of
 * attention.py:536-549): dq2[r, h*d+dd] = sum_c g[h,r,c] Wk[h*d+dd, c] and dWk[h*d+dd, c] = sum_r q2[r, h*d+dd] g[h,r,c].
 * g [heads][R][C] bf16, q2 / dq2 [R][C] bf16, wk = the bf16 rows of Wk (row stride wk_ld), dwk [C][C] fp32 dense.
 * bf16, d = 64, C = 768, R % 32 == 0 (focus_traj_time2_gw_ok); ws: focus_traj_time2_gw_workspace_bytes. */
int focus_traj_time2_gw_ok(int R, int heads, int d, int dtype);
size_t focus_traj_time2_gw_workspace_bytes(int R, int heads, int d);
int focus_traj_time2_gw(const void* g, const void* q2, const void* wk, int64_t wk_ld, void* dq2, float* dwk, void* ws,
                        size_t ws_bytes, int R, int heads, int d, int dtype, void* stream);

/* Row passes over the [frames * tokens, vocabulary] tensors of STEVE.forward (gumbel.hip); one workgroup per row, the row in
 * registers: V % 8 == 0, 8 <= V <= 8192, R < 2^31, dtype fp32 or bf16 (focus_rows_ok).
 *
 * focus_gumbel_fwd (steve.py:262-271 with utils.py:47-61): logp = log_softmax(x);  z = softmax((logp + G_soft) / tau), or the
 * straight-through value (one_hot(argmax z) - z) + z when `hard`;  target[r] = argmax(logp + G_hard) -- the only use the
 * reference makes of its second, hard sample (target may be NULL).  G = -log(E + tiny), E ~ Exp(1): e_soft / e_hard [R, V]
 * fp32 are the draws (the reference's torch.empty_like(logits).exponential_()), or both NULL and the draws are generated in
 * the kernel from `seed` (4 x uint32 on the device: soft stream, hard stream), which the backward regenerates.
 * x, dx [R, V] `dtype`; z, dz [R, V] `dtype_z` (= dtype, or bf16 under fp32 logits: a bf16 decoder behind an fp32 encoder);
 * stats [R, 4] fp32 (row max, log-sum, and the same of the tempered logits) for the backward.
 * focus_gumbel_bwd: dx = d(loss)/d(x) from dz = d(loss)/d(z) (gradient of the soft sample in both modes). */
int focus_rows_ok(int64_t R, int V, int dtype);
int focus_gumbel_fwd(const void* x, const float* e_soft, const float* e_hard, const void* seed, void* z, int64_t* target,
                     float* stats, int64_t R, int V, float tau, int hard, int dtype, int dtype_z, void* stream);
int focus_gumbel_bwd(const void* x, const float* e_soft, const void* seed, const float* stats, const void* dz, void* dx,
                     int64_t R, int V, float tau, int dtype, int dtype_z, void* stream);

/* Label-smoothing cross entropy (losses.py:53-59; steve.py:303-306 with smoothing 0) on fp32 or bf16 logits [R, V]:
 * loss_rows [R] and the row log-sum-exp lse [R] fp32; the backward rebuilds softmax from lse and writes
 * dlogits = ((softmax - (1 - smoothing) onehot - smoothing / V) / R) * g[0] in `dtype` (g: device scalar, fp32). */
int focus_xent_rows_fwd(const void* logits, const int64_t* target, float* loss_rows, float* lse, int64_t R, int V,
                        float smoothing, int dtype, void* stream);
int focus_xent_rows_bwd(const void* logits, const int64_t* target, const float* lse, const float* g, void* dlogits,
                        int64_t R, int V, float smoothing, int dtype, void* stream);

/* out[i] = (res ? res[i] : 0) + keep(i) * y[i] * 65536 / (65536 - thr): a residual branch that ends in nn.Dropout
 * (transformer.py:45-47, :147-163) in one pass.  keep(i) = 16 hashed bits of (seed, i) >= thr, thr = round(p * 65536);
 * seed: 2 x uint32 on the device.  The branch's backward is the same call on the incoming gradient with res = NULL.
 * n % 8 == 0, 16-byte aligned pointers, fp32 or bf16. */
int focus_dropout_add(const void* y, const void* res, const void* seed, int thr, void* out, int64_t n, int dtype,
                      void* stream);

/* xdiag[b,s,:] = xt[b,s,s/P,:] (attention.py:533-535) and its adjoint dxt[b,s,s/P,:] += dxdiag[b,s,:]. */
int focus_diag_gather(const void* xt, void* xdiag, int B, int S, int F, int C, int dtype, void* stream);
int focus_diag_scatter_add(const void* dxdiag, void* dxt, int B, int S, int F, int C, int dtype, void* stream);

/* out[b, i] = (x ? x[b, i] : 0) + s[b] * y[b, i]   for b < B, i < per  (per % 8 == 0).
 * Stochastic depth on a residual branch in one pass (common.py:46-60: x + drop_path(y)) and its adjoint (dy = s * dout
 * with x == NULL).  keep == 0: s = scale.  keep > 0: scale holds the U[0,1) draws and s[b] = floor(keep + scale[b]) / keep
 * (the reference's mask / keep_prob, formed in the kernel instead of by three elementwise launches). */
int focus_scale_add(const void* x, const void* y, const float* scale, float keep, void* out, int B, int64_t per,
                    int dtype, void* stream);

/* dtype conversion (weights shadow copies, gradient casts): n elements. */
int focus_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FOCUS_AMD_H */
