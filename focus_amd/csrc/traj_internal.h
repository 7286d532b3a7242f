#pragma once
#include <hip/hip_runtime.h>
bool focus_traj_space_mfma_ok(int P, int d, int heads, int dtype);
int focus_traj_space_fwd_mfma(const void* qkv, void* xt, void* xdiag, float* lse, int B, int F, int P, int heads,
                              hipStream_t s);
