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

#ifdef __HIPCC__
// Workgroups that share blockIdx.y (one (batch, head)) read the same K / V / Q panels.  The dispatcher deals consecutive
// linear workgroup ids round-robin over the 8 XCDs, so the gridDim.x workgroups of one (b, h) land on up to 8 different L2s
// and each fetches the panels from HBM again (PMC: traj_dq 506 MB per launch against ~215 MB algorithmic).  This remap gives
// every XCD a contiguous range of the linear ids instead: all workgroups of a (b, h) on one XCD, panels fetched once.
// Placement is a speed matter only (MI355X_MICROARCH.md, Workgroup dispatch): any assignment computes the same result.
__device__ __forceinline__ void focus_xcd_group(int& bx, int& by) {
    const int gx = gridDim.x, total = gx * gridDim.y;
    const int id = blockIdx.x + gx * blockIdx.y;
    const int xcd = id & 7, slot = id >> 3;
    const int q = total >> 3, r = total & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    by = v / gx;
    bx = v - by * gx;
}
#endif
