#pragma once
#include <hip/hip_runtime.h>
bool focus_traj_space_mfma_ok(int P, int d, int heads, int dtype);
int focus_traj_space_fwd_mfma(const void* qkv, void* xt, void* xdiag, float* lse, int B, int F, int P, int heads,
                              hipStream_t s);
int focus_traj_space_bwd_mfma(const void* qkv, const void* xt, const float* lse, const void* dxt, const void* dxdiag,
                              float* delta, void* dqkv, int B, int F, int P, int heads, hipStream_t s);
