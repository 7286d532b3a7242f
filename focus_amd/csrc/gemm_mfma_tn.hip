// gemm_mfma_tn.hip -- bf16 MFMA GEMM for operands that are BOTH strided along the reduction index:
//     C[I,J] (+)= alpha * sum_m  P[m, I]^T . Q[m, J]          P, Q row-major, m = their row index
// This is the weight-gradient product dW[N,K] = dY[M,N]^T . X[M,K] computed straight from the row-major
// activations -- no transposed copies of dY and X (the NT kernel needs both re-laid K-contiguous, which cost two
// extra HBM round trips per Linear per step).
// LDS tiles keep the global layout ([64 rows of m][128 columns], 256-B rows, DMA'd by global_load_lds_dwordx4
// with a 16-B chunk swizzle (swz below) applied on the source address); the MFMA fragments, whose k index
// runs over tile ROWS, are gathered with ds_read_b64_tr_b16 (cdna_hip_programming.md T10; semantics pinned by
// tests/test_gpu_parity.py::test_tr16_probe...).  128x128 output tile, 4 waves (2x2), v_mfma_f32_16x16x32_bf16,
// double-buffered; the reduction over m is split over workgroups (the output is tiny, the reduction long): partial
// tiles go to per-split fp32 slabs summed by a reduce kernel (or, without a workspace, fp32 atomics into a zeroed C).
// Large outputs with long reductions take the wave-specialised kernel of gemm_mfma_tn_ws.hip; this one serves the rest.
#include "focus_common.h"
#include "gemm_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BI = 128, BJ = 128, BKM = 64;        // output tile, reduction rows per step
constexpr int TILE = BKM * 256;                    // 16 KiB per operand per stage

union Frag { bf16x8 v; s16x4 t[2]; };

// byte offset of element (row, col) [col multiple of 4] in a [64][128 bf16] tile, 16-B chunks swizzled by row
// The swizzle is chosen for the transposed reads: one 32-lane bank group of ds_read_b64_tr_b16 touches tile rows
// {r..r+3} u {r+8..r+11} x two adjacent chunks, so the XOR key must give those 8 rows 8 different EVEN values
// (key = row & 15 put rows r and r+1 on the same chunk pair: a 2-way conflict on every read, measured as
// SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).
__device__ __forceinline__ int swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }
__device__ __forceinline__ int toff(int row, int col) {
    return row * 256 + ((((col >> 3)) ^ swz(row)) << 4) + (col & 4) * 2;
}

// k = tile rows r0..r0+7 (natural order), m/n = tile column c0 + (lane & 15); lane group g = lane>>4 selects r0.
__device__ __forceinline__ bf16x8 col_frag16(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15;
    const int row = r0 + (i >> 2), col = c0 + 4 * (i & 3);
    Frag f;
    f.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + toff(row, col)));
    f.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + toff(row + 4, col)));
    return f.v;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const focus_gemm_desc d, int tiles_i, int tiles_j, int splits,
                                                         int m_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][P tile | Q tile]
    const int nwg = tiles_i * tiles_j * splits;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // split-major order: the workgroups an XCD runs together work on the SAME slice of the reduction and on an
    // 8-row-tile-deep patch of the output, so the P and Q rows of that slice are fetched into its L2 once and shared
    const int tiles = tiles_i * tiles_j;
    const int split = lid / tiles, tile = lid % tiles;
    constexpr int GM = 8;
    const int group = tile / (GM * tiles_j), first_i = group * GM;
    const int gsz = min(tiles_i - first_i, GM), in_g = tile - group * GM * tiles_j;
    const int i0 = (first_i + in_g % gsz) * BI, j0 = (in_g / gsz) * BJ;
    const int Mred = d.K;                                   // reduction length (rows of P and Q)
    const int m_begin = split * m_per_split;
    const int m_end = min(Mred, m_begin + m_per_split);
    const int nk = (m_end - m_begin + BKM - 1) / BKM;
    if (nk <= 0) return;

    // batch (blockIdx.y, batch1 only): independent products whose outputs are stacked densely ([batch][M][N])
    const int bz = blockIdx.y, nbz = gridDim.y;
    const bf16_t* Pm = static_cast<const bf16_t*>(d.A) + bz * d.bsA1;     // P[m][i] : element (i, m) of "A" = Pm[m*ldp + i]
    const bf16_t* Qm = static_cast<const bf16_t*>(d.B) + bz * d.bsB1;     // Q[m][j]
    const int64_t ldp = d.csA, ldq = d.rsB;
    // slab mode: this split's partial tile goes to aux[split][batch][M][N] with plain stores (summed by tn_reduce_kernel)
    float* C = d.aux ? static_cast<float*>(d.aux) + ((int64_t)split * nbz + bz) * d.M * d.N : static_cast<float*>(d.C);
    const int64_t ldc = d.aux ? d.N : d.rsC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w >> 1, wj = w & 1;

    // DMA: one wave instruction = 4 tile rows x 256 B; lane -> (row_in = lane>>4, cpos = lane&15).
    // wave w, instruction g (0..3) covers tile rows (w*4 + g)*4 .. +3 ; rows/chunks are clamped into the matrix.
    const int cpos = lane & 15, rin = lane >> 4;
    auto stage = [&](int st, int kt) __attribute__((always_inline)) {
        char* sp = smem + st * 2 * TILE;
        char* sq = sp + TILE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = (w * 4 + g) * 4 + rin;
            const int m = min(m_begin + kt * BKM + row, Mred - 1);
            const int ch = cpos ^ swz(row);
            const int ci = min(i0 + ch * 8, d.M - 8), cj = min(j0 + ch * 8, d.N - 8);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(Pm + (int64_t)m * ldp + ci), (lvoid_t*)(sp + (w * 4 + g) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(Qm + (int64_t)m * ldq + cj), (lvoid_t*)(sq + (w * 4 + g) * 1024), 16, 0, 0);
        }
    };
    // rows past the end of the reduction were DMA'd from a clamped address: zero them in the P tile
    auto zero_tail = [&](int st, int kt) __attribute__((always_inline)) {
        const int valid = m_end - (m_begin + kt * BKM);      // rows of this step that are real
        if (valid >= BKM) return;
        char* sp = smem + st * 2 * TILE;
        for (int e = tid; e < BKM * 16; e += 256) {
            const int row = e >> 4;
            if (row >= valid) *reinterpret_cast<uint4*>(sp + row * 256 + (e & 15) * 16) = make_uint4(0, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fq = lane >> 4;
    auto compute = [&](int st) __attribute__((always_inline)) {
        const char* sp = smem + st * 2 * TILE;
        const char* sq = sp + TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fp[4], fqv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fp[a] = col_frag16(sp, ks * 32 + 8 * fq, wi * 64 + a * 16, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) fqv[b] = col_frag16(sq, ks * 32 + 8 * fq, wj * 64 + b * 16, lane);
            // D[j][i] = sum_m Q[m][j] * P[m][i]: lane ends up with 4 consecutive j of one i
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqv[b], fp[a], acc[a][b], 0, 0, 0);
        }
    };

    // only the last K-step of the last split can hold rows past the end of the reduction
    const bool has_tail = (m_end - m_begin) % BKM != 0;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (has_tail && nk == 1) { zero_tail(0, 0); __syncthreads(); }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        stage(1, kt + 1);
        compute(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (has_tail && kt + 2 == nk) { zero_tail(1, kt + 1); __syncthreads(); }
        if (kt + 2 < nk) stage(0, kt + 2);
        compute(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (has_tail && kt + 3 == nk) { zero_tail(0, kt + 2); __syncthreads(); }
    }
    if (kt < nk) compute(0);

    // acc[a][b][r4] = D[j = j0 + wj*64 + b*16 + fq*4 + r4][i = i0 + wi*64 + a*16 + (lane&15)]
    const int fr = lane & 15;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gi = i0 + wi * 64 + a * 16 + fr;
        if (gi >= d.M) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = j0 + wj * 64 + b * 16 + fq * 4;
            if (d.aux) {
                if (gj < d.N)      // N % 8 == 0: the 4 columns are in range together
                    *reinterpret_cast<float4*>(C + (int64_t)gi * ldc + gj) =
                        make_float4(d.alpha * acc[a][b][0], d.alpha * acc[a][b][1], d.alpha * acc[a][b][2], d.alpha * acc[a][b][3]);
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    if (gj + r4 < d.N) atomicAdd(C + (int64_t)gi * ldc + (gj + r4), d.alpha * acc[a][b][r4]);
            }
        }
    }
}

__global__ void tn_reduce_many_kernel(const float* __restrict__ parts, float* __restrict__ out, int64_t n4, int splits, int N,
                                      int64_t rsC);
// Sum of the split slabs, few splits (large outputs): one float4 per thread, the splits walked in order.
__global__ void tn_reduce_kernel(const float* __restrict__ parts, float* __restrict__ out, int64_t n4, int splits,
                                 int N, int64_t rsC) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4*>(parts)[i];
#pragma unroll 4                                                      // (four slab loads in flight: left rolled, each waits for the last)
    for (int k = 1; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(parts)[(int64_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t e = i * 4, row = e / N, col = e % N;
    *reinterpret_cast<float4*>(out + row * rsC + col) = s;
}

static void launch_tn_reduce(const float* parts, float* out, int64_t n4, int splits, int N, int64_t rsC, hipStream_t s) {
    if (splits >= 16)
        hipLaunchKernelGGL(tn_reduce_many_kernel, dim3((unsigned)cdiv64(n4, 16)), dim3(256), 0, s, parts, out, n4, splits, N, rsC);
    else
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, s, parts, out, n4, splits, N, rsC);
}

// Many splits (small outputs).  16 float4 columns x 16 split lanes per block: a small output with many splits (192 x 192 from
// 128 slabs -- the per-frame k/v projection gradients of STEVE) used to be 36 blocks walking 128 slabs serially
// (31 us, as long as the GEMM itself); the split lanes issue their loads together and meet in LDS.
__global__ __launch_bounds__(256) void tn_reduce_many_kernel(const float* __restrict__ parts, float* __restrict__ out, int64_t n4,
                                                             int splits, int N, int64_t rsC) {
    __shared__ float4 red[16][16];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + c;                     // one float4 column per 16 threads
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        int k = sl;
        for (; k + 48 < splits; k += 64) {                               // four loads in flight per thread
            const float4 a = reinterpret_cast<const float4*>(parts)[(int64_t)k * n4 + i];
            const float4 b = reinterpret_cast<const float4*>(parts)[(int64_t)(k + 16) * n4 + i];
            const float4 cc = reinterpret_cast<const float4*>(parts)[(int64_t)(k + 32) * n4 + i];
            const float4 d = reinterpret_cast<const float4*>(parts)[(int64_t)(k + 48) * n4 + i];
            s.x += (a.x + b.x) + (cc.x + d.x); s.y += (a.y + b.y) + (cc.y + d.y);
            s.z += (a.z + b.z) + (cc.z + d.z); s.w += (a.w + b.w) + (cc.w + d.w);
        }
        for (; k < splits; k += 16) {
            const float4 v = reinterpret_cast<const float4*>(parts)[(int64_t)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[sl][c] = s;
    __syncthreads();
    if (sl == 0 && i < n4) {
#pragma unroll
        for (int r = 1; r < 16; ++r) { const float4 v = red[r][c]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        const int64_t e = i * 4, row = e / N, col = e % N;
        *reinterpret_cast<float4*>(out + row * rsC + col) = s;
    }
}

// out[n] = sum_r parts[r][n] for a short, wide slab (bias-gradient partials: tens of rows, N columns).  tn_reduce_kernel
// would walk the rows serially in N/4 threads (latency-bound: 84 dependent round trips measured 20+ us); here 16 row
// lanes share a float4 column and meet in LDS.
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ parts, float* __restrict__ out, int n4, int rows) {
    __shared__ float4 red[16][16];
    const int c = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n4)
        for (int r = rr; r < rows; r += 16) {
            const float4 v = reinterpret_cast<const float4*>(parts)[(int64_t)r * n4 + col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[rr][c] = s;
    __syncthreads();
    if (rr == 0 && col < n4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][c]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(out)[col] = s;
    }
}

// both reductions of focus_linear_wgrad in one launch: blocks [0, nb_main) sum the dW slabs, the rest the bias partials
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ parts, float* __restrict__ out, int64_t n4,
                                                           int splits, int N, int64_t rsC, int nb_main,
                                                           const float* __restrict__ bparts, float* __restrict__ bout, int bn4,
                                                           int brows) {
    if ((int)blockIdx.x < nb_main) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n4) return;
        float4 s = reinterpret_cast<const float4*>(parts)[i];
#pragma unroll 4
        for (int k = 1; k < splits; ++k) {
            const float4 v = reinterpret_cast<const float4*>(parts)[(int64_t)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int64_t e = i * 4, row = e / N, col = e % N;
        *reinterpret_cast<float4*>(out + row * rsC + col) = s;
        return;
    }
    __shared__ float4 red[16][16];
    const int c = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const int col = ((int)blockIdx.x - nb_main) * 16 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < bn4)
#pragma unroll 4
        for (int r = rr; r < brows; r += 16) {
            const float4 v = reinterpret_cast<const float4*>(bparts)[(int64_t)r * bn4 + col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[rr][c] = s;
    __syncthreads();
    if (rr == 0 && col < bn4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][c]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(bout)[col] = s;
    }
}

struct TnPlan { int tiles_i, tiles_j, splits, m_per_split; };
TnPlan tn_plan(int M, int N, int K, int batch = 1) {
    TnPlan p;
    p.tiles_i = (M + BI - 1) / BI;
    p.tiles_j = (N + BJ - 1) / BJ;
    const int tiles = p.tiles_i * p.tiles_j * batch;
    p.splits = std::max(1, std::min((2 * 256 + tiles - 1) / tiles, (K + 4 * BKM - 1) / (4 * BKM)));
    if (K <= 1024) p.splits = 1;        // short reductions (slot / object-token layers): one unit per tile, no reduce launch
    p.m_per_split = ((K + p.splits - 1) / p.splits + BKM - 1) / BKM * BKM;
    p.splits = (K + p.m_per_split - 1) / p.m_per_split;
    return p;
}

}  // namespace

extern "C" size_t focus_gemm_tn_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const focus_tn_plan ws = focus_gemm_tn_ws_plan(M, N, K);
    if (ws.kind) return (size_t)ws.splits * M * N * sizeof(float);
    return (size_t)tn_plan(M, N, K).splits * M * N * sizeof(float);
}

extern "C" size_t focus_gemm_tn_batched_workspace_bytes(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    if (batch == 1) return focus_gemm_tn_workspace_bytes(M, N, K);
    return (size_t)tn_plan(M, N, K, batch).splits * batch * M * N * sizeof(float);
}

// A described as [M, K] with rsA == 1 (P[m][i] row-major: csA = ld of P), B as [K, N] with csB == 1 (Q[m][j]).
bool focus_gemm_mfma_tn_ok(const focus_gemm_desc& d) {
    static const bool enabled = !(getenv("FOCUS_GEMM_TN") && atoi(getenv("FOCUS_GEMM_TN")) == 0);
    if (!enabled || d.dtype_ab != FOCUS_BF16 || d.dtype_c != FOCUS_F32) return false;
    if (d.rsA != 1 || d.csB != 1 || d.csC != 1) return false;
    if (d.bias || d.residual || d.epilogue != FOCUS_EPI_NONE) return false;
    if (d.batch0 * d.batch1 != 1) {
        // batched form (batch1 only): slab mode, outputs stacked densely [batch][M][N]
        if (d.batch0 != 1 || !d.aux || d.rsC != d.N || d.bsC1 != (int64_t)d.M * d.N || (d.bsA1 & 7) || (d.bsB1 & 7) ||
            d.batch1 > 65535)
            return false;
    }
    if (!d.aux && !d.accumulate) return false;          // atomic mode needs a zero-initialised accumulate target
    if (d.aux && (d.accumulate || (d.rsC & 3) || !focus_aligned(d.aux, 16) || !focus_aligned(d.C, 16))) return false;
    if ((d.csA & 7) || (d.rsB & 7) || (d.M & 7) || (d.N & 7) || d.M < 8 || d.N < 8 || d.K < 1) return false;
    if (!focus_aligned(d.A, 16) || !focus_aligned(d.B, 16)) return false;
    return true;
}

int focus_gemm_mfma_tn(const focus_gemm_desc& d, hipStream_t s) {
    if (!focus_gemm_mfma_tn_ok(d)) return FOCUS_ERR_ALIGN;
    const int nb = d.batch0 * d.batch1;
    focus_tn_plan ws = focus_gemm_tn_ws_plan(d.M, d.N, d.K);
    if (nb > 1) ws.kind = 0;                         // batched products take the uniform kernel
    if (ws.kind) {                                   // large outputs with a long reduction: wave-specialised kernel
        const int rc = focus_gemm_mfma_tn_ws(d, ws, nullptr, s);
        if (rc != FOCUS_OK) return rc;
        if (d.aux) {
            const int64_t n4 = (int64_t)d.M * d.N / 4;
            launch_tn_reduce((const float*)d.aux, (float*)d.C, n4, ws.splits, d.N, d.rsC, s);
            FOCUS_CHECK_LAUNCH();
        }
        return FOCUS_OK;
    }
    const TnPlan pl = tn_plan(d.M, d.N, d.K, nb);
    const int tiles_i = pl.tiles_i, tiles_j = pl.tiles_j, tiles = tiles_i * tiles_j, splits = pl.splits;
    const int m_per_split = pl.m_per_split;
    const size_t lds = 4 * TILE;
    static bool once = (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    // one split and a dense output: the "slab" is C itself, no reduce launch
    const bool direct = d.aux && splits == 1 && d.rsC == d.N;
    focus_gemm_desc dd = d;
    if (direct) dd.aux = d.C;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits, nb), dim3(256), lds, s, dd, tiles_i, tiles_j, splits, m_per_split);
    FOCUS_CHECK_LAUNCH();
    if (d.aux && !direct) {
        const int64_t n4 = (int64_t)nb * d.M * d.N / 4;
        launch_tn_reduce((const float*)d.aux, (float*)d.C, n4, splits, d.N, d.rsC, s);
        FOCUS_CHECK_LAUNCH();
    }
    return FOCUS_OK;
}

// ---- nn.Linear weight + bias gradient in one pass over dY ----------------------------------------------
extern "C" size_t focus_linear_wgrad_workspace_bytes(int N, int K, int M) {
    if (N <= 0 || K <= 0 || M <= 0) return 0;
    size_t bytes = focus_gemm_tn_workspace_bytes(N, K, M);
    const focus_tn_plan ws = focus_gemm_tn_ws_plan(N, K, M);
    if (ws.kind) bytes += (size_t)ws.splits * ws.tiles_j * N * sizeof(float);      // column-sum partials
    return bytes;
}

extern "C" int focus_linear_wgrad(const void* dy, const void* x, float* dw, float* db, void* ws, size_t ws_bytes,
                                  int M, int N, int K, int64_t ld_dy, int64_t ld_x, int dtype, void* stream) {
    if (!dy || !x || !dw || !ws) return FOCUS_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return FOCUS_ERR_SHAPE;
    if (dtype != FOCUS_BF16) return FOCUS_ERR_DTYPE;
    if (ws_bytes < focus_linear_wgrad_workspace_bytes(N, K, M)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    focus_gemm_desc d = {};
    d.M = N; d.N = K; d.K = M; d.batch0 = 1; d.batch1 = 1;
    d.A = dy; d.rsA = 1; d.csA = ld_dy;          // A[i, m] = dy[m, i]
    d.B = x; d.rsB = ld_x; d.csB = 1;            // B[m, j] = x[m, j]
    d.C = dw; d.rsC = K; d.csC = 1;
    d.aux = ws; d.alpha = 1.f; d.dtype_ab = FOCUS_BF16; d.dtype_c = FOCUS_F32;
    if (!focus_gemm_mfma_tn_ok(d)) return FOCUS_ERR_ALIGN;
    const focus_tn_plan pl = focus_gemm_tn_ws_plan(N, K, M);
    if (pl.kind && db) {
        float* csum = static_cast<float*>(ws) + (size_t)pl.splits * N * K;
        int rc = focus_gemm_mfma_tn_ws(d, pl, csum, s);
        if (rc != FOCUS_OK) return rc;
        const int64_t n4 = (int64_t)N * K / 4;
        const int nb_main = (int)cdiv64(n4, 256), nb_bias = (int)cdiv64(N / 4, 16);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb_main + nb_bias), dim3(256), 0, s, (const float*)ws, dw, n4, pl.splits,
                           K, (int64_t)K, nb_main, (const float*)csum, db, N / 4, pl.splits * pl.tiles_j);
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    int rc = focus_gemm_mfma_tn(d, s);
    if (rc != FOCUS_OK) return rc;
    if (db) return focus_colsum(dy, db, M, N, ld_dy, 0, dtype, stream);
    return FOCUS_OK;
}
