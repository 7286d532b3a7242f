// traj_space_bwd_mfma.hip -- fused backward of the space step of trajectory attention (bf16, head dim 64).
//
// Flash-style: nothing of size S x S is stored.  With P[s,f,p] = exp(scale*q_s.k_{f,p} - lse[s,f]) recomputed from
// the forward's log-sum-exp, dX = dx~ (+ dx_diag on the query's own frame), delta[s,f] = dX[s,f,:].x~[s,f,:]:
//     dV_{f,p} = sum_s P * dX[s,f,:]          dP = dX[s,f,:].v_{f,p}       dL = P * (dP - delta) * scale
//     dK_{f,p} = sum_s dL * q_s               dQ_s = sum_{f,p} dL * k_{f,p}
// Three kernels, no atomics, every output written exactly once:
//   traj_delta_kernel : delta [B,h,S,F]                                   (HBM-bound, reads dx~ and x~ once)
//   traj_dq_kernel    : workgroup = 128 queries of one (b,h), walks the frames; keys on the accumulator rows
//                       (swapped products), dL feeds  dQ^T += K^T.dL  straight from the accumulator registers;
//                       K^T fragments come from the row-major K tile through ds_read_b64_tr_b16.
//   traj_dkv_kernel   : workgroup = the <=224 keys of one (b,h,frame), one wave per 32 keys, walks all queries;
//                       queries on the accumulator rows, P and dL feed  dV += P^T.dX,  dK += dL^T.Q  as A operands,
//                       dX / Q column fragments through ds_read_b64_tr_b16.
#include "focus_common.h"
#include "traj_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int HD = 64;
constexpr int QT = 128;
constexpr float LOG2E = 1.44269504088896341f;

union Pack8 { bf16x8 v; bf16_t e[8]; uint4 u; uint2 h2[2]; s16x4 t[2]; };

// 16-B chunk swizzle of a [rows][64 bf16] tile with 128-B rows.  A row covers half of the 64 LDS banks (row parity
// picks the half), so the XOR key is built from row>>1, bit-reversed: (a) a ds_read_b128 bank group reads one chunk
// of 16 rows whose row>>1 values are 8 distinct ones mod 8 -> 16 distinct (half, chunk) slots; (b) a
// ds_read_b64_tr_b16 bank group reads 4 adjacent chunks of rows r..r+3 -> rows r and r+2 need keys that differ in
// bit 2, which the reversal provides.  (key = row & 7 was 2-way conflicted on both: SQ_LDS_BANK_CONFLICT ~40 %.)
__device__ __forceinline__ int swz(int row, int chunk) {
    const int key = ((row & 2) << 1) | ((row >> 1) & 2) | ((row >> 3) & 1);
    return row * 128 + ((chunk ^ key) << 4);
}

// MFMA 32x32x16 operand whose k index runs over tile ROWS and whose m/n index is a tile COLUMN, in the k order of
// an accumulator tile used as the other operand: element j <-> row r0 + 8*(j>>2) + (j&3), column c0 + (lane&31),
// where the caller passes r0 = 16*s + 4*(lane>>5) (+ block base).  Two transposed LDS reads (T10).
__device__ __forceinline__ bf16x8 col_frag(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    const int row = r0 + (i >> 2);
    Pack8 p;
    p.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + swz(row, col >> 3) + (col & 4) * 2));
    p.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + swz(row + 8, col >> 3) + (col & 4) * 2));
    return p.v;
}

__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s2) {
    Pack8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p.e[j] = f32_to_bf16(a[8 * s2 + j]);
    return p.v;
}

// ------------------------------------------------------------------------------------------------
// delta[b,h,s,f] = sum_d (dxt[b,s,f,h,d] + [f == s/P] dxdiag[b,s,h,d]) * xt[b,s,f,h,d]
// one wave per (b,s); a head's 64 channels sit on 16 adjacent lanes (4 channels per lane)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void traj_delta_kernel(const bf16_t* __restrict__ dxt, const bf16_t* __restrict__ dxdiag,
                                                         const bf16_t* __restrict__ xt, float* __restrict__ delta,
                                                         int64_t rows, int S, int F, int P, int heads) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C = heads * HD;
    const int64_t b = row / S;
    const int s = (int)(row % S), fs = s / P;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + lane * 4;
        const bool act = c < C;
        f4 dd = {0.f, 0.f, 0.f, 0.f};
        if (act) dd = ld4<bf16_t>(dxdiag + row * C + c);
        for (int f = 0; f < F; ++f) {
            float p = 0.f;
            if (act) {
                f4 g = ld4<bf16_t>(dxt + (row * F + f) * C + c);
                const f4 x = ld4<bf16_t>(xt + (row * F + f) * C + c);
                if (f == fs) { g.x += dd.x; g.y += dd.y; g.z += dd.z; g.w += dd.w; }
                p = g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            if (act && (lane & 15) == 0) delta[((b * heads + c / HD) * S + s) * F + f] = p;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ
// ------------------------------------------------------------------------------------------------
template <int NKB>
__global__ __launch_bounds__(256, 2) void traj_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dxt,
                                                         const bf16_t* __restrict__ dxdiag, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                         int B, int F, int P, int heads) {
    constexpr int KROWS = NKB * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                       // [KROWS][64] bf16 row-major, chunk-swizzled
    char* sV = smem + KROWS * 128;         // same for V
    char* slabs = smem + 2 * KROWS * 128;  // 4 x [32][128 B]

    const int S = F * P, N = S + 1, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const int bh = blockIdx.y, b = bh / heads, hh = bh % heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int s_raw = blockIdx.x * QT + w * 32 + r;
    const bool q_valid = s_raw < S;
    const int s_q = min(s_raw, S - 1), fs = s_q / P;
    const bf16_t* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;

    bf16x8 qf[4];
    Pack8 dd[4];   // dx_diag row fragments (added to dx~ on the query's own frame)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)(1 + s_q) * tok + ks * 16 + 8 * h);
        dd[ks].u = *reinterpret_cast<const uint4*>(dxdiag + ((int64_t)b * S + s_q) * C + hh * HD + ks * 16 + 8 * h);
    }
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

    for (int f = 0; f < F; ++f) {
        __syncthreads();
        for (int e = tid; e < KROWS * 8; e += 256) {
            const int p = e >> 3, c = e & 7;
            uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
            if (p < P) {
                const bf16_t* row = base + (int64_t)(1 + f * P + p) * tok + c * 8;
                kv = *reinterpret_cast<const uint4*>(row + C);
                vv = *reinterpret_cast<const uint4*>(row + 2 * C);
            }
            *reinterpret_cast<uint4*>(sK + swz(p, c)) = kv;
            *reinterpret_cast<uint4*>(sV + swz(p, c)) = vv;
        }
        // dX fragments of this lane's query for frame f (B operand: [k=d][col=q])
        bf16x8 df[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Pack8 g;
            g.u = *reinterpret_cast<const uint4*>(dxt + (((int64_t)b * S + s_q) * F + f) * C + hh * HD + ks * 16 + 8 * h);
            if (f == fs) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g.e[j] = f32_to_bf16(bf16_to_f32(g.e[j]) + bf16_to_f32(dd[ks].e[j]));
            }
            df[ks] = g.v;
        }
        const int64_t sf = (((int64_t)b * heads + hh) * S + s_q) * F + f;
        const float lse2 = q_valid ? lse[sf] * LOG2E : INFINITY;
        const float del = delta[sf];
        __syncthreads();

#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 sa, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + swz(kb * 32 + r, ks * 2 + h));
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + swz(kb * 32 + r, ks * 2 + h));
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], dp, 0, 0, 0);
            }
            // padded keys need no mask here: their K rows are staged as zeros, so whatever dL they get is multiplied
            // by K^T = 0 in the dQ product (and exp2(0 - lse2) is finite)
            const float dels = del * scale;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(sa[i], c2, -lse2));
                sa[i] = pr * fmaf(dp[i], scale, -dels);   // dL[key][q] = P * (dP - delta) * scale
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 lf = pack_acc(sa, s2);
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) {
                    const bf16x8 kt = col_frag(sK, kb * 32 + 16 * s2 + 4 * h, dblk * 32, lane);   // K^T[d][key]
                    dq[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, lf, dq[dblk], 0, 0, 0);
                }
            }
        }
    }
    // dQ^T[d][q] -> rows of dqkv through the wave's LDS slab (same scheme as the forward)
    __syncthreads();
    char* slab = slabs + w * 4096;
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(dq[dblk][4 * g + 0]) | ((uint32_t)f32_to_bf16(dq[dblk][4 * g + 1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(dq[dblk][4 * g + 2]) | ((uint32_t)f32_to_bf16(dq[dblk][4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(slab + r * 128 + (((dblk * 8 + 2 * g + h) ^ (r & 15)) << 3)) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
        const int row = p4 * 8 + (lane >> 3), q8 = lane & 7;
        uint4 raw = *reinterpret_cast<const uint4*>(slab + row * 128 + ((q8 ^ ((row & 15) >> 1)) << 4));
        if (row & 1) { const uint32_t a0 = raw.x, a1 = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = a0; raw.w = a1; }
        const int s_row = blockIdx.x * QT + w * 32 + row;
        if (s_row < S)
            *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + 1 + s_row) * tok + hh * HD + q8 * 8) = raw;
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV
// ------------------------------------------------------------------------------------------------
constexpr int QC = 32;   // queries per chunk

template <int NKB>
__global__ __launch_bounds__(64 * NKB) void traj_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dxt,
                                                            const bf16_t* __restrict__ dxdiag,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int B, int F, int P, int heads) {
    constexpr int NT = 64 * NKB;
    __shared__ __attribute__((aligned(16))) char sQ[2][QC * 128];
    __shared__ __attribute__((aligned(16))) char sD[2][QC * 128];
    __shared__ __attribute__((aligned(16))) float sLse[2][QC];
    __shared__ __attribute__((aligned(16))) float sDel[2][QC];

    const int S = F * P, N = S + 1, C = heads * HD;
    const int64_t tok = 3 * (int64_t)C;
    const int f = blockIdx.x, bh = blockIdx.y, b = bh / heads, hh = bh % heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* base = qkv + (int64_t)b * N * tok + hh * HD;
    const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;

    // this wave's 32 keys as B operands [k=d][col=key]: kept in registers for the whole sweep
    const int key = w * 32 + r;
    const bool key_ok = key < P;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        Pack8 a, c;
        a.u = make_uint4(0, 0, 0, 0); c.u = a.u;
        if (key_ok) {
            const bf16_t* row = base + (int64_t)(1 + f * P + key) * tok + ks * 16 + 8 * h;
            a.u = *reinterpret_cast<const uint4*>(row + C);
            c.u = *reinterpret_cast<const uint4*>(row + 2 * C);
        }
        kf[ks] = a.v; vf[ks] = c.v;
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

    const int nchunk = (S + QC - 1) / QC;
    auto stage = [&](int buf, int chunk) __attribute__((always_inline)) {
        for (int e = tid; e < 2 * QC * 8; e += NT) {
            const int which = e / (QC * 8), idx = e % (QC * 8), row = idx >> 3, c = idx & 7;
            const int s = chunk * QC + row;
            Pack8 v; v.u = make_uint4(0, 0, 0, 0);
            if (s < S) {
                if (which == 0) {
                    v.u = *reinterpret_cast<const uint4*>(base + (int64_t)(1 + s) * tok + c * 8);
                } else {
                    v.u = *reinterpret_cast<const uint4*>(dxt + (((int64_t)b * S + s) * F + f) * C + hh * HD + c * 8);
                    if (s / P == f) {
                        Pack8 d2;
                        d2.u = *reinterpret_cast<const uint4*>(dxdiag + ((int64_t)b * S + s) * C + hh * HD + c * 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v.e[j] = f32_to_bf16(bf16_to_f32(v.e[j]) + bf16_to_f32(d2.e[j]));
                    }
                }
            }
            *reinterpret_cast<uint4*>((which == 0 ? sQ[buf] : sD[buf]) + swz(row, c)) = v.u;
        }
        for (int e = tid; e < QC; e += NT) {
            const int s = chunk * QC + e;
            const int64_t sf = (((int64_t)b * heads + hh) * S + min(s, S - 1)) * F + f;
            sLse[buf][e] = s < S ? lse[sf] * LOG2E : INFINITY;      // +inf -> P = 0 for padded queries
            sDel[buf][e] = s < S ? delta[sf] : 0.f;
        }
    };

    stage(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) stage(buf ^ 1, ch + 1);
        const char* tq = sQ[buf];
        const char* td = sD[buf];
        // S'[q][key] and dP'[q][key]: queries on the accumulator rows, this wave's keys on the lanes
        f32x16 sa, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(tq + swz(r, ks * 2 + h));
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(td + swz(r, ks * 2 + h));
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        // rows of the tile: q = (i&3) + 8*(i>>2) + 4*h  -> 4 consecutive floats per group of 4 registers
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(&sLse[buf][8 * g + 4 * h]);
            const float4 d4 = *reinterpret_cast<const float4*>(&sDel[buf][8 * g + 4 * h]);
            const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                const float pr = key_ok ? __builtin_amdgcn_exp2f(fmaf(sa[i], c2, -ls[e])) : 0.f;
                sa[i] = pr;                                   // P[q][key]
                dp[i] = pr * fmaf(dp[i], scale, -de[e] * scale);   // dL[q][key]
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack_acc(sa, s2), lf = pack_acc(dp, s2);
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk) {
                const bf16x8 dxc = col_frag(td, 16 * s2 + 4 * h, dblk * 32, lane);   // dX[q][d] by columns
                const bf16x8 qc = col_frag(tq, 16 * s2 + 4 * h, dblk * 32, lane);    // Q[q][d] by columns
                dv[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, dxc, dv[dblk], 0, 0, 0);
                dk[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf, qc, dk[dblk], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // dK[key][d], dV[key][d]: accumulator row = key (in-block), column (lane) = d
#pragma unroll
    for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kl = w * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (kl < P) {
                bf16_t* row = dqkv + ((int64_t)b * N + 1 + f * P + kl) * tok + hh * HD + dblk * 32 + r;
                row[C] = f32_to_bf16(dk[dblk][i]);
                row[2 * C] = f32_to_bf16(dv[dblk][i]);
            }
        }
}

template <int NKB>
int launch_bwd(const void* qkv, const void* dxt, const void* dxdiag, const float* lse, const float* delta, void* dqkv,
               int B, int F, int P, int heads, hipStream_t s) {
    const int S = F * P;
    {
        const size_t lds = (size_t)2 * NKB * 32 * 128 + 4 * 4096;
        auto k = traj_dq_kernel<NKB>;
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
        (void)once;
        hipLaunchKernelGGL(k, dim3((S + QT - 1) / QT, B * heads), dim3(256), lds, s, (const bf16_t*)qkv, (const bf16_t*)dxt,
                           (const bf16_t*)dxdiag, lse, delta, (bf16_t*)dqkv, B, F, P, heads);
        FOCUS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL((traj_dkv_kernel<NKB>), dim3(F, B * heads), dim3(64 * NKB), 0, s, (const bf16_t*)qkv,
                       (const bf16_t*)dxt, (const bf16_t*)dxdiag, lse, delta, (bf16_t*)dqkv, B, F, P, heads);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

// Patch-token rows of dqkv (q, k and v parts of tokens 1..N-1) are fully written; the cls row/parts are the caller's.
// delta: [B,h,S,F] fp32 scratch.
int focus_traj_space_bwd_mfma(const void* qkv, const void* xt, const float* lse, const void* dxt, const void* dxdiag,
                              float* delta, void* dqkv, int B, int F, int P, int heads, hipStream_t s) {
    if (B * heads > 65535) return FOCUS_ERR_SHAPE;
    const int S = F * P;
    const int64_t rows = (int64_t)B * S;
    hipLaunchKernelGGL(traj_delta_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, (const bf16_t*)dxt,
                       (const bf16_t*)dxdiag, (const bf16_t*)xt, delta, rows, S, F, P, heads);
    FOCUS_CHECK_LAUNCH();
    switch ((P + 31) / 32) {
        case 1: return launch_bwd<1>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
        case 2: return launch_bwd<2>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
        case 3: return launch_bwd<3>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
        case 4: return launch_bwd<4>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
        case 5: return launch_bwd<5>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
        case 6: return launch_bwd<6>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
        default: break;
    }
    return launch_bwd<7>(qkv, dxt, dxdiag, lse, delta, dqkv, B, F, P, heads, s);
}
