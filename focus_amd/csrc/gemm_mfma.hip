// gemm_mfma.hip -- bf16 MFMA GEMM for gfx950:  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)  (fp32 accumulate)
//
// Both operands are contiguous along K (the nn.Linear forward layout: activations [M,K], weight [N,K]).
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Global->register->LDS staging with the next K-tile's loads issued before
// the current tile's MFMAs (cdna_hip_programming.md T14), double-buffered LDS, one barrier per K-step,
// XOR-swizzled 128-byte LDS rows so the ds_read_b128 fragment reads spread over 8 slots (T2).
// The MFMA is issued "swapped" (weight fragment as A, activation fragment as B) so each lane ends up
// with 4 consecutive output columns of one row: 8-byte (bf16) / 16-byte (fp32) stores.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous band of M-tiles (T1).
#include "focus_common.h"
#include "gemm_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <typename TC>
__device__ __forceinline__ float epi_one(const focus_gemm_desc& d, float v, int n, const TC* X, int64_t off) {
    v *= d.alpha;
    if (d.bias) v += d.bias[n];
    switch (d.epilogue) {
        case FOCUS_EPI_GELU: v = gelu_erf(v); break;  // aux (pre-activation) handled by the caller
        case FOCUS_EPI_RELU: v = fmaxf(v, 0.f); break;
        case FOCUS_EPI_TANH: v = tanhf(v); break;
        case FOCUS_EPI_DGELU: v *= dgelu_erf(ld<TC>(X + off)); break;
        case FOCUS_EPI_DRELU: v = ld<TC>(X + off) > 0.f ? v : 0.f; break;
        case FOCUS_EPI_DTANH: { float y = ld<TC>(X + off); v *= (1.f - y * y); } break;
        default: break;
    }
    return v;
}

template <typename TC>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const focus_gemm_desc d, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]

    // ---- XCD-aware tile assignment (bijective for any grid size) ----
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.y;
    const int b0 = z / d.batch1, b1 = z % d.batch1;
    const bf16_t* A = static_cast<const bf16_t*>(d.A) + b0 * d.bsA0 + b1 * d.bsA1;
    const bf16_t* B = static_cast<const bf16_t*>(d.B) + b0 * d.bsB0 + b1 * d.bsB1;
    const int64_t coff = b0 * d.bsC0 + b1 * d.bsC1;
    TC* C = static_cast<TC*>(d.C) + coff;
    const TC* R = d.residual ? static_cast<const TC*>(d.residual) + coff : nullptr;
    TC* X = d.aux ? static_cast<TC*>(d.aux) + coff : nullptr;
    const int64_t lda = d.rsA, ldb = d.csB;  // B is described as [K,N]: csB = stride between its N rows

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;

    // staging map: chunk e = tid + i*256 -> (row = e>>3, chunk = e&7); 8 threads cover one 128-B row
    uint4 ra[4], rb[4];
    auto g_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256, row = e >> 3, c = e & 7;
            const int gm = m0 + row, gn = n0 + row;
            const int64_t ko = (int64_t)kt * BK + c * 8;
            ra[i] = gm < d.M ? *reinterpret_cast<const uint4*>(A + gm * lda + ko) : make_uint4(0, 0, 0, 0);
            rb[i] = gn < d.N ? *reinterpret_cast<const uint4*>(B + gn * ldb + ko) : make_uint4(0, 0, 0, 0);
        }
    };
    auto s_store = [&](int stage) {
        char* sa = smem + stage * 2 * TILE_BYTES;
        char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256, row = e >> 3, c = e & 7;
            *reinterpret_cast<uint4*>(sa + swz(row, c)) = ra[i];
            *reinterpret_cast<uint4*>(sb + swz(row, c)) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = d.K / BK;
    g_load(0);
    s_store(0);
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) g_load(kt + 1);
        const char* sa = smem + stage * 2 * TILE_BYTES;
        const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + frow;
                fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz(row, ks * 4 + fq));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + frow;
                fb[j] = *reinterpret_cast<const bf16x8*>(sb + swz(row, ks * 4 + fq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) s_store(stage ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][r] = D[n = n0+wn*64+j*16+fq*4+r][m = m0+wm*64+i*16+frow] ----
    const size_t va = 4 * sizeof(TC);
    const bool vec_ok = (d.csC == 1) && ((d.rsC & 3) == 0) && (reinterpret_cast<uintptr_t>(C) % va == 0) &&
                        (!R || reinterpret_cast<uintptr_t>(R) % va == 0) &&
                        (!X || reinterpret_cast<uintptr_t>(X) % va == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + wm * 64 + i * 16 + frow;
        if (gm >= d.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + wn * 64 + j * 16 + fq * 4;
            if (gn >= d.N) continue;
            const int64_t off = gm * d.rsC + (int64_t)gn * d.csC;
            float v[4];
            if (vec_ok && gn + 3 < d.N) {
                if (d.epilogue == FOCUS_EPI_GELU && X) {
                    f4 pre;
                    pre.x = d.alpha * acc[i][j][0] + (d.bias ? d.bias[gn] : 0.f);
                    pre.y = d.alpha * acc[i][j][1] + (d.bias ? d.bias[gn + 1] : 0.f);
                    pre.z = d.alpha * acc[i][j][2] + (d.bias ? d.bias[gn + 2] : 0.f);
                    pre.w = d.alpha * acc[i][j][3] + (d.bias ? d.bias[gn + 3] : 0.f);
                    st4<TC>(X + off, pre);
                }
                f4 xa = {0.f, 0.f, 0.f, 0.f};
                if (d.epilogue >= FOCUS_EPI_DGELU) xa = ld4<TC>(X + off);
                const float xs[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    float t = d.alpha * acc[i][j][r4];
                    if (d.bias) t += d.bias[gn + r4];
                    switch (d.epilogue) {
                        case FOCUS_EPI_GELU: t = gelu_erf(t); break;
                        case FOCUS_EPI_RELU: t = fmaxf(t, 0.f); break;
                        case FOCUS_EPI_TANH: t = tanhf(t); break;
                        case FOCUS_EPI_DGELU: t *= dgelu_erf(xs[r4]); break;
                        case FOCUS_EPI_DRELU: t = xs[r4] > 0.f ? t : 0.f; break;
                        case FOCUS_EPI_DTANH: t *= (1.f - xs[r4] * xs[r4]); break;
                        default: break;
                    }
                    v[r4] = t;
                }
                if (R) { f4 rr = ld4<TC>(R + off); v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
                if (d.accumulate) { f4 cc = ld4<TC>(C + off); v[0] += cc.x; v[1] += cc.y; v[2] += cc.z; v[3] += cc.w; }
                st4<TC>(C + off, (f4){v[0], v[1], v[2], v[3]});
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    if (gn + r4 >= d.N) break;
                    const int64_t o = off + (int64_t)r4 * d.csC;
                    if (d.epilogue == FOCUS_EPI_GELU && X)
                        st<TC>(X + o, d.alpha * acc[i][j][r4] + (d.bias ? d.bias[gn + r4] : 0.f));
                    float t = epi_one<TC>(d, acc[i][j][r4], gn + r4, X, o);
                    if (R) t += ld<TC>(R + o);
                    if (d.accumulate) t += ld<TC>(C + o);
                    st<TC>(C + o, t);
                }
            }
        }
    }
}

}  // namespace

bool focus_gemm_mfma_nt_ok(const focus_gemm_desc& d) {
    if (d.dtype_ab != FOCUS_BF16) return false;
    if (d.csA != 1 || d.rsB != 1) return false;             // both contiguous along K
    if (d.K <= 0 || (d.K % BK) != 0) return false;
    if ((d.rsA & 7) || (d.csB & 7) || (d.bsA0 & 7) || (d.bsA1 & 7) || (d.bsB0 & 7) || (d.bsB1 & 7)) return false;
    if (!focus_aligned(d.A, 16) || !focus_aligned(d.B, 16)) return false;
    if (d.M < 1 || d.N < 1) return false;
    return true;
}

int focus_gemm_mfma_nt(const focus_gemm_desc& d, hipStream_t s) {
    if (!focus_gemm_mfma_nt_ok(d)) return FOCUS_ERR_ALIGN;
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    const int nbatch = d.batch0 * d.batch1;
    if (nbatch > 65535) return FOCUS_ERR_SHAPE;
    dim3 grid(tiles_m * tiles_n, nbatch);
    const size_t lds = 4 * TILE_BYTES;
    if (d.dtype_c == FOCUS_BF16)
        hipLaunchKernelGGL((gemm_nt_kernel<bf16_t>), grid, dim3(256), lds, s, d, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<float>), grid, dim3(256), lds, s, d, tiles_m, tiles_n);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

// ---- public dispatcher ---------------------------------------------------------------------------
extern "C" int focus_gemm(const focus_gemm_desc* desc, void* stream) {
    if (!desc || !desc->A || !desc->B || !desc->C) return FOCUS_ERR_NULL;
    focus_gemm_desc d = *desc;
    if (d.batch0 < 1) d.batch0 = 1;
    if (d.batch1 < 1) d.batch1 = 1;
    if (d.M <= 0 || d.N <= 0) return FOCUS_OK;
    if (d.K < 0) return FOCUS_ERR_SHAPE;
    if (d.accumulate && d.dtype_c != FOCUS_F32) return FOCUS_ERR_DTYPE;
    if (d.epilogue >= FOCUS_EPI_DGELU && !d.aux) return FOCUS_ERR_NULL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (focus_gemm_mfma_nt_ok(d)) return focus_gemm_mfma_nt(d, s);
    return focus_gemm_generic(d, s);
}

extern "C" int focus_linear_fwd(const void* x, const void* w, const float* bias, const void* residual, void* y,
                                void* aux, int M, int N, int K, int epilogue, int dtype, void* stream) {
    focus_gemm_desc d = {};
    d.M = M; d.N = N; d.K = K; d.batch0 = 1; d.batch1 = 1;
    d.A = x; d.rsA = K; d.csA = 1;
    d.B = w; d.rsB = 1; d.csB = K;      // B[k,n] = w[n,k]
    d.C = y; d.rsC = N; d.csC = 1;
    d.bias = bias; d.residual = residual; d.aux = aux;
    d.alpha = 1.f; d.accumulate = 0; d.epilogue = epilogue; d.dtype_ab = dtype; d.dtype_c = dtype;
    return focus_gemm(&d, stream);
}
