// traj_time2_gw.hip -- the two consumers of g = d(loss)/d(u) of the re-associated temporal step (traj_time2.hip;
// attention.py:536-549 backward) in ONE pass over g:
//     dq2[r, h*64 + dd] = sum_c g[h, r, c] * Wk[h*64 + dd, c]                 (per head: [R x 768] . [768 x 64])
//     dWk[h*64 + dd, c] = sum_r q2[r, h*64 + dd] * g[h, r, c]                 (per head: [64 x R] . [R x 768])
// g is [heads][R][C] bf16 -- 12x the size of dq2 (231 MB at the bench shape).  As two batched GEMM launches each of them
// streamed it once (HBM-bound: 86 + 80 us per block, 170-190 TF/s, 2.1 ms per step); here a workgroup owns a row range of
// one head and uses every 32-row tile of g for both products while it sits in LDS.
//   * Wk[h] [64 x 768] lives in REGISTERS as MFMA B fragments (24 per wave: wave w owns the 16-row block w & 1 of the tile
//     and the 16 columns (w >> 1) of the head), the dq2 product reads its A fragments (g rows) with ds_read_b128;
//   * dWk[h] [64 x 768] fp32 accumulates in registers over the whole row range (24 accumulator fragments per wave: all 64
//     dd x the 96 channels 96 w .. 96 w + 95); both operands of that product have their reduction index (the row) along
//     the tile rows, so they are gathered with ds_read_b64_tr_b16 (T10) -- from the SAME g image the row reads use (the
//     dual-use image (b) of cdna_hip_programming.md T10: 256-byte rows, chunk ^= ((row & 3) << 2) | ((row >> 2) & 3));
//   * g and q2 tiles arrive by LDS-DMA, double buffered (the next tile streams in during the current tile's 48 MFMAs);
//   * per-range dWk partials go to fp32 slabs [splits][C][C], summed by time2_gw_reduce_kernel.
#include "focus_common.h"
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int HD = 64, TR = 32;                       // head dim, rows per tile
constexpr int C_ = 768;                               // channels (12 heads x 64): the instance built
constexpr int NSUB = C_ / 128;                        // [32][128] sub-tiles of 256-byte rows per g tile
constexpr int GT_BYTES = TR * C_ * 2, QT_BYTES = TR * HD * 2, STAGE = GT_BYTES + QT_BYTES;

union Frag { bf16x8 v; s16x4 t[2]; uint4 u; };

// byte offset of 16-byte chunk ch (0..15) of row `row` inside a [32][128]-column sub-tile (256-byte rows)
__device__ __forceinline__ int offb(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// MFMA 16x16x32 operand whose k index runs over TILE ROWS r0 .. r0+7 and whose m/n index is the column c0 + (lane & 15)
// of a 128-column sub-tile: two transposed reads (lane 4q+p of a 16-lane group supplies row q, columns 4p .. 4p+3)
__device__ __forceinline__ bf16x8 tr_frag_g(const char* sub, int r0, int c0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int ch = (c0 >> 3) + (p >> 1), bo = 8 * (p & 1);
    Frag f;
    f.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sub + offb(r0 + q, ch) + bo));
    f.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sub + offb(r0 + 4 + q, ch) + bo));
    return f.v;
}
// the same from the q2 tile [32 rows][64 dd] (128-byte rows, plain layout)
__device__ __forceinline__ bf16x8 tr_frag_q(const char* qt, int r0, int c0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int o = (c0 + 4 * p) * 2;
    Frag f;
    f.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(qt + (r0 + q) * 128 + o));
    f.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(qt + (r0 + 4 + q) * 128 + o));
    return f.v;
}

__global__ __launch_bounds__(512) void time2_gw_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ q2,
                                                       const bf16_t* __restrict__ wk, int64_t wk_ld, bf16_t* __restrict__ dq2,
                                                       float* __restrict__ slabs, int R, int heads, int tiles_per_split) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];            // 2 stages x (g tile | q2 tile)
    const int h = blockIdx.y, sp = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles_all = R / TR;
    const int t_begin = sp * tiles_per_split, t_end = min(ntiles_all, t_begin + tiles_per_split);
    const int row16_ = lane & 15, kq_ = lane >> 4;
    const int mb = w & 1, nbq = w >> 1;                                      // dq2: tile rows 16 mb.., head columns 16 nbq..
    const bf16_t* gh = g + (int64_t)h * R * C_;

    // Wk[h] rows 16 nbq + (lane & 15), all 768 k: 24 B fragments in registers
    bf16x8 wf[C_ / 32];
    {
        const bf16_t* wrow = wk + (int64_t)(h * HD + nbq * 16 + row16_) * wk_ld + kq_ * 8;
#pragma unroll
        for (int ks = 0; ks < C_ / 32; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    }
    f32x4 accw[4][6];                                                        // dWk[16 mbk + ..][96 w + 16 j + ..]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 6; ++j) accw[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the weight loads are done before any DMA is counted
#pragma unroll
    for (int ks = 0; ks < C_ / 32; ++ks) asm volatile("" : "+v"(wf[ks]));

    // DMA of tile t into stage st: 48 g pieces (4 rows x 256 B) + 4 q2 pieces (8 rows x 128 B), dealt over the 8 waves
    const uint32_t sm_a = lds_addr_of(smem);
    auto dma = [&](int t, int st) __attribute__((always_inline)) {
        const int64_t r0 = (int64_t)t * TR;
        const uint32_t base = sm_a + st * STAGE;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int p = w * 6 + j, sub = p >> 3, rg = p & 7;               // sub-tile, 4-row group
            const int row = rg * 4 + (lane >> 4), pos = lane & 15;
            const int ch = pos ^ (((row & 3) << 2) | ((row >> 2) & 3));       // the chunk this position keeps
            glds16(gh + (r0 + row) * C_ + sub * 128 + ch * 8, __builtin_amdgcn_readfirstlane(base + sub * 8192 + rg * 1024));
        }
        if (w < 4) {
            const int row = w * 8 + (lane >> 3);
            glds16(q2 + (r0 + row) * (int64_t)(heads * HD) + h * HD + (lane & 7) * 8,
                   __builtin_amdgcn_readfirstlane(base + GT_BYTES + w * 1024));
        }
    };
    if (t_begin >= t_end) return;
    dma(t_begin, 0);
    int st = 0;
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) dma(t + 1, st ^ 1);
        // tile t landed: this wave's pieces of t+1 (6, or 7 for the q2 waves) may stay in flight
        if (more) { if (w < 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                        // every wave's pieces of tile t are in LDS
        const char* gt = smem + st * STAGE;
        const char* qt = gt + GT_BYTES;
        // the ~45 LDS offsets below are loop invariant; hoisted out of the loop they cost 45 VGPRs next to the 192 of the weight
        // fragments and the accumulators (spills, whose reloads drain the DMA in flight): recomputed per tile from an opaque copy
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int row16 = lane_o & 15, kq = lane_o >> 4;
        // ---- dq2 tile: rows 16 mb .., columns 16 nbq .. of the head ----
        f32x4 accq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C_ / 32; ++ks) {
            const int col = ks * 32 + kq * 8;                                // 8 consecutive channels of row 16 mb + row16
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(gt + (col >> 7) * 8192 + offb(mb * 16 + row16, (col & 127) >> 3));
            accq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ks], accq, 0, 0, 0);
        }
        {
            bf16_t* o = dq2 + ((int64_t)t * TR + mb * 16 + kq * 4) * (heads * HD) + h * HD + nbq * 16 + row16;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(int64_t)r * (heads * HD)] = f32_to_bf16(accq[r]);
        }
        // ---- dWk += q2^T . g over the 32 rows of the tile ----
        // (the q2 fragments are re-read per channel block instead of held: 254 VGPRs + spills otherwise)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int c = w * 96 + j * 16;
            const bf16x8 gb = tr_frag_g(gt + (c >> 7) * 8192, 8 * kq, c & 127, lane_o);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bf16x8 qa = tr_frag_q(qt, 8 * kq, a * 16, lane_o);
                accw[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, gb, accw[a][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                        // everyone is done with stage st: tile t+2 may land there
        st ^= 1;
    }
    // partial dWk[h]: accw[a][j][r] = D[dd = 16 a + 4 kq + r][c = 96 w + 16 j + (lane & 15)]
    float* out = slabs + ((int64_t)sp * heads * HD + h * HD) * C_;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(int64_t)(a * 16 + kq_ * 4 + r) * C_ + w * 96 + j * 16 + row16_] = accw[a][j][r];
}

__global__ __launch_bounds__(256) void time2_gw_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dwk, int64_t n4,
                                                              int splits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4*>(slabs)[i];
#pragma unroll 4
    for (int k = 1; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(slabs)[(int64_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dwk)[i] = s;
}

int gw_splits(int R, int heads) {
    const int tiles = R / TR;
    return std::max(1, std::min(tiles, 256 / std::max(heads, 1)));
}

}  // namespace

// bf16, head dim 64, C = heads * 64 = 768, R % 32 == 0
extern "C" int focus_traj_time2_gw_ok(int R, int heads, int d, int dtype) {
    return dtype == FOCUS_BF16 && d == HD && heads * d == C_ && R >= TR && R % TR == 0;
}

extern "C" size_t focus_traj_time2_gw_workspace_bytes(int R, int heads, int d) {
    if (!focus_traj_time2_gw_ok(R, heads, d, FOCUS_BF16)) return 0;
    return (size_t)gw_splits(R, heads) * C_ * C_ * sizeof(float);
}

// g [heads][R][C] bf16 (head-major), q2 [R][C] bf16, wk: bf16 rows h*64+dd of Wk (row stride wk_ld elements);
// dq2 [R][C] bf16 out; dwk [C][C] fp32 out (dense); ws: focus_traj_time2_gw_workspace_bytes.
extern "C" int focus_traj_time2_gw(const void* g, const void* q2, const void* wk, int64_t wk_ld, void* dq2, float* dwk, void* ws,
                                   size_t ws_bytes, int R, int heads, int d, int dtype, void* stream) {
    if (!g || !q2 || !wk || !dq2 || !dwk || !ws) return FOCUS_ERR_NULL;
    if (!focus_traj_time2_gw_ok(R, heads, d, dtype) || (wk_ld & 7)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(g, 16) || !focus_aligned(q2, 16) || !focus_aligned(wk, 16) || !focus_aligned(dwk, 16)) return FOCUS_ERR_ALIGN;
    if (ws_bytes < focus_traj_time2_gw_workspace_bytes(R, heads, d)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int splits = gw_splits(R, heads), tiles = R / TR;
    const int tps = (tiles + splits - 1) / splits;
    const int used = (tiles + tps - 1) / tps;                       // splits that actually own tiles
    auto k = time2_gw_kernel;
    const size_t lds = 2 * (size_t)STAGE;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(k, dim3(used, heads), dim3(512), lds, s, (const bf16_t*)g, (const bf16_t*)q2, (const bf16_t*)wk, wk_ld,
                       (bf16_t*)dq2, (float*)ws, R, heads, tps);
    FOCUS_CHECK_LAUNCH();
    const int64_t n4 = (int64_t)C_ * C_ / 4;
    hipLaunchKernelGGL(time2_gw_reduce_kernel, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, s, (const float*)ws, dwk, n4, used);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
