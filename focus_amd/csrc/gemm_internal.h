// gemm_internal.h -- internal entry points behind focus_gemm().
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/focus_amd.h"

int focus_gemm_generic(const focus_gemm_desc& d, hipStream_t s);
// bf16 MFMA kernel for K-contiguous A [M,K] and B given as [N,K] rows (rsB==1): returns
// FOCUS_ERR_ALIGN / FOCUS_ERR_SHAPE when the operands do not qualify (caller falls back).
int focus_gemm_mfma_nt(const focus_gemm_desc& d, hipStream_t s);
bool focus_gemm_mfma_nt_ok(const focus_gemm_desc& d);
// bf16 MFMA kernel for operands strided along the reduction (dW = dY^T . X from row-major activations).
bool focus_gemm_mfma_tn_ok(const focus_gemm_desc& d);
int focus_gemm_mfma_tn(const focus_gemm_desc& d, hipStream_t s);
// wave-specialised (4 consumer + 4 loader waves, 3-stage ring) NT kernel for bf16 output with aligned rows
bool focus_gemm_mfma_ws_ok(const focus_gemm_desc& d);
int focus_gemm_mfma_ws(const focus_gemm_desc& d, hipStream_t s);
// wave-specialised TN kernel (8 consumer + 4 loader waves); the plan also sizes the slab workspace
struct focus_tn_plan { int kind, tiles_i, tiles_j, splits, m_per_split; };
focus_tn_plan focus_gemm_tn_ws_plan(int M, int N, int K);
int focus_gemm_mfma_tn_ws(const focus_gemm_desc& d, const focus_tn_plan& pl, float* csum, hipStream_t s);
// small row counts (M <= 1024, K <= 1536): one 32x32 tile per workgroup, in-workgroup split-K, operands in registers
bool focus_gemm_mfma_small_ok(const focus_gemm_desc& d);
int focus_gemm_mfma_small(const focus_gemm_desc& d, hipStream_t s);
