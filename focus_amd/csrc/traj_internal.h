#pragma once
#include <hip/hip_runtime.h>
// fused space-attention kernels: up to 14 blocks of 32 keys per frame (P <= 448: the HR 16x336 grid of 21x21 patches
// + objects); the dQ kernel is instantiated per block count
#define FOCUS_TRAJ_MAX_KEY_BLOCKS 14
bool focus_traj_space_mfma_ok(int P, int d, int heads, int dtype);
void focus_traj_space_tiling(int P, int* nkb, int* nt);
int focus_traj_space_fwd_mfma(const void* qkv, void* xt, void* xdiag, float* lse, int B, int F, int P, int heads,
                              hipStream_t s);
int focus_traj_space_bwd_mfma(const void* qkv, const void* xt, const float* lse, const void* dxt, const void* dxdiag,
                              float* delta, float* lse2, void* dxsum, void* dqkv, int B, int F, int P, int heads,
                              hipStream_t s);
// cls row (one query per (b,h) over all N keys); bwd ADDS onto rows 1.. of the k/v parts of dqkv, writes row 0.
bool focus_traj_cls_ok(int N, int d);
size_t focus_traj_cls_scratch_floats(int B, int N, int heads);
int focus_traj_cls_fwd(const void* qkv, void* cls_out, float* cls_lse, float* scratch, int B, int N, int heads, int dtype,
                       hipStream_t s);
int focus_traj_cls_bwd(const void* qkv, const float* cls_lse, const void* dcls, void* dqkv, float* scratch, int B, int N,
                       int heads, int dtype, hipStream_t s);   // scratch: 2*B*heads*N floats
// re-associated temporal step (traj_time2.hip)
bool focus_traj_time2_ok(int F, int heads, int d, int dtype);
