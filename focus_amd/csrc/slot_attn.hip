// slot_attn.hip -- STEVE slot-attention inverted softmax + weighted mean (steve.py:76-83), fwd and bwd.
// One launch streams k_t and v_t exactly once: per input row n the wave computes the K slot logits
// (K wave reductions over D), the softmax over slots, writes attn_vis and accumulates the un-normalised
// update sum_n (attn+eps)[n,k] * v[n,:] in registers (K x D/64 per lane).  A finish kernel reduces the
// per-chunk partials and applies the 1/sum_n normalisation.  HBM-bound: 2*N*D*esize read per (b,t,iter).
#include "focus_common.h"
#include <cstdlib>
#include <utility>

namespace {

constexpr int ROWS_PER_BLOCK = 64;    // 16 rows per wave: 8 waves/SIMD in flight (256 left the loads latency-bound: 404 us fwd)

template <typename T, int KP, int DV>
__global__ __launch_bounds__(256) void slot_fwd_kernel(const T* __restrict__ kt, const T* __restrict__ vt,
                                                       int64_t kv_bs, const T* __restrict__ q,
                                                       T* __restrict__ attn, int64_t attn_bs,
                                                       float* __restrict__ partial, int N, int K, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // q [K][D], then reduce area
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < K * D; i += 256) sm[i] = ld<T>(q + (int64_t)b * K * D + i);
    __syncthreads();
    float U[KP][DV], cs[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        cs[k] = 0.f;
#pragma unroll
        for (int j = 0; j < DV; ++j) U[k][j] = 0.f;
    }
    const int n_end = min(N, (chunk + 1) * ROWS_PER_BLOCK);
    for (int n = chunk * ROWS_PER_BLOCK + w; n < n_end; n += 4) {
        const T* kr = kt + (int64_t)b * kv_bs + (int64_t)n * D;
        const T* vr = vt + (int64_t)b * kv_bs + (int64_t)n * D;
        float kv[DV], vv[DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int e = j * 64 + lane;
            kv[j] = e < D ? ld<T>(kr + e) : 0.f;
            vv[j] = e < D ? ld<T>(vr + e) : 0.f;
        }
        float lg[KP];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            float p = 0.f;
            if (k < K) {
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) p += kv[j] * sm[k * D + e];
                }
            }
            lg[k] = k < K ? wave_sum(p) : -INFINITY;
            m = fmaxf(m, lg[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) { lg[k] = k < K ? __expf(lg[k] - m) : 0.f; den += lg[k]; }
        const float inv = 1.f / den;
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float a = lg[k] * inv;
            if (k == lane) mine = a;
            if (k < K) {
                const float ae = a + eps;
                cs[k] += ae;
#pragma unroll
                for (int j = 0; j < DV; ++j) U[k][j] += ae * vv[j];
            }
        }
        if (lane < K) st<T>(attn + (int64_t)b * attn_bs + (int64_t)n * K + lane, mine);
    }
    // combine the 4 waves, write partial[b][chunk][k][0..D-1], colsum at [..][D]
    float* red = sm + K * D;  // [4][64]
    float* out = partial + ((int64_t)b * nchunk + chunk) * K * (D + 1);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K)
#pragma unroll
        for (int j = 0; j < DV + 1; ++j) {
            const float val = j < DV ? U[k][j < DV ? j : 0] : cs[k];
            __syncthreads();
            red[w * 64 + lane] = val;
            __syncthreads();
            if (w == 0) {
                const float t = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
                if (j < DV) { const int e = j * 64 + lane; if (e < D) out[k * (D + 1) + e] = t; }
                else if (lane == 0) out[k * (D + 1) + D] = t;  // every lane of a wave holds that wave's row sum
            }
        }
    }
}

// (An in-launch finish -- the workgroup that publishes a batch element's last partial reduces all 16, the "split-K in one
// launch" ticket protocol -- was measured and dropped: with agent-scope release fences every one of the 512 workgroups
// writes the L2 back, 22 -> 68 us per launch; with write-through (sc1) partial stores and no fences the single reducing
// workgroup reads its 135 KB through dependent sc1 round trips, 22 -> 51 us.  The separate launch below costs 6 us.)
// grid (K, B), 64 channel lanes x FIN_CL chunk lanes: reduce the partials over the chunks, normalise.  (With 64
// threads walking the chunks one after the other the launch was a chain of dependent L2 round trips: 16.6 us for 16
// chunks; here every thread's loads are independent and the four chunk lanes meet in LDS.)
constexpr int FIN_CL = 16;      // chunk lanes of the finish kernels: 64 channel lanes x 16 chunk lanes = 1024 threads, so
                                // that with <= 16 chunks every load of the launch is issued at once (one round trip)
template <typename T>
__global__ __launch_bounds__(64 * FIN_CL) void slot_fwd_finish(const float* __restrict__ partial, T* __restrict__ upd,
                                                               float* __restrict__ colsum, int nchunk, int K, int D) {
    __shared__ float red[FIN_CL][4][64], redc[FIN_CL];
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const float* p = partial + (int64_t)b * nchunk * K * (D + 1) + (int64_t)k * (D + 1);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, c = 0.f;
    for (int ch = cl; ch < nchunk; ch += FIN_CL) {
        const float* pc = p + (int64_t)ch * K * (D + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < D) s[j] += pc[lane + 64 * j];
        if (lane == 0) c += pc[D];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[cl][j][lane] = s[j];
    if (lane == 0) redc[cl] = c;
    __syncthreads();
    if (cl < 4) {                       // wave j of the first four sums channel block j over the chunk lanes
        const int j = cl;
        c = 0.f;
#pragma unroll
        for (int q = 0; q < FIN_CL; ++q) c += redc[q];
        if (j == 0 && lane == 0) colsum[b * K + k] = c;
        if (lane + 64 * j < D) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < FIN_CL; ++q) t += red[q][j][lane];
            st<T>(upd + ((int64_t)b * K + k) * D + lane + 64 * j, t / c);
        }
    }
}

template <typename T, int KP, int DV>
__global__ __launch_bounds__(256) void slot_bwd_kernel(const T* __restrict__ kt, const T* __restrict__ vt,
                                                       int64_t kv_bs, const T* __restrict__ q,
                                                       const T* __restrict__ attn, int64_t attn_bs,
                                                       const float* __restrict__ colsum, const T* __restrict__ upd,
                                                       const T* __restrict__ dupd, const T* __restrict__ dattn,
                                                       T* __restrict__ dkt, T* __restrict__ dvt, int accumulate,
                                                       float* __restrict__ partial, int N, int K, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // q [K][D] | dupd [K][D] | r[K] | cs[K] | red
    float* sq = sm;
    float* sdu = sm + K * D;
    float* sr = sdu + K * D;
    float* scs = sr + 32;
    float* red = scs + 32;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < K * D; i += 256) {
        sq[i] = ld<T>(q + (int64_t)b * K * D + i);
        sdu[i] = ld<T>(dupd + (int64_t)b * K * D + i);
    }
    __syncthreads();
    // r[k] = dupd[k,:].upd[k,:]  (one wave per slot, round robin)
    for (int k = w; k < K; k += 4) {
        float p = 0.f;
        for (int e = lane; e < D; e += 64) p += sdu[k * D + e] * ld<T>(upd + ((int64_t)b * K + k) * D + e);
        p = wave_sum(p);
        if (lane == 0) { sr[k] = p; scs[k] = colsum[b * K + k]; }
    }
    __syncthreads();
    float dQ[KP][DV];
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
        for (int j = 0; j < DV; ++j) dQ[k][j] = 0.f;
    const int n_end = min(N, (chunk + 1) * ROWS_PER_BLOCK);
    for (int n = chunk * ROWS_PER_BLOCK + w; n < n_end; n += 4) {
        const int64_t ro = (int64_t)b * kv_bs + (int64_t)n * D;
        float kv[DV], vv[DV], dv[DV], dk[DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int e = j * 64 + lane;
            kv[j] = e < D ? ld<T>(kt + ro + e) : 0.f;
            vv[j] = e < D ? ld<T>(vt + ro + e) : 0.f;
            dv[j] = 0.f; dk[j] = 0.f;
        }
        float av[KP], dav[KP];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            av[k] = 0.f; dav[k] = 0.f;
            if (k < K) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) p += sdu[k * D + e] * vv[j];
                }
                const float dw = wave_sum(p);
                av[k] = ld<T>(attn + (int64_t)b * attn_bs + (int64_t)n * K + k);
                const float wgt = (av[k] + eps) / scs[k];
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) dv[j] += wgt * sdu[k * D + e];
                }
                dav[k] = (dw - sr[k]) / scs[k];
                if (dattn) dav[k] += ld<T>(dattn + (int64_t)b * attn_bs + (int64_t)n * K + k);
                dot += av[k] * dav[k];
            }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (k < K) {
                const float dl = av[k] * (dav[k] - dot);
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int e = j * 64 + lane;
                    if (e < D) { dk[j] += dl * sq[k * D + e]; dQ[k][j] += dl * kv[j]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int e = j * 64 + lane;
            if (e < D) {
                float a = dk[j], c = dv[j];
                if (accumulate) { a += ld<T>(dkt + ro + e); c += ld<T>(dvt + ro + e); }
                st<T>(dkt + ro + e, a);
                st<T>(dvt + ro + e, c);
            }
        }
    }
    float* out = partial + ((int64_t)b * nchunk + chunk) * K * D;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K)
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            __syncthreads();
            red[w * 64 + lane] = dQ[k][j];
            __syncthreads();
            if (w == 0) {
                const int e = j * 64 + lane;
                if (e < D) out[k * D + e] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64 * FIN_CL) void slot_bwd_finish(const float* __restrict__ partial, T* __restrict__ dq, int nchunk,
                                                               int K, int D) {
    __shared__ float red[FIN_CL][4][64];
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = cl; ch < nchunk; ch += FIN_CL) {
        const float* pc = partial + (((int64_t)b * nchunk + ch) * K + k) * D;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < D) s[j] += pc[lane + 64 * j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[cl][j][lane] = s[j];
    __syncthreads();
    if (cl < 4 && lane + 64 * cl < D) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < FIN_CL; ++q) t += red[q][cl][lane];
        st<T>(dq + ((int64_t)b * K + k) * D + lane + 64 * cl, t);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16, K <= 16 slots, D % 32 == 0, D <= 256: slot logits (and, backward, dupd . v) on the matrix pipe.
// A wave owns 64 input rows.  Phase 1: four v_mfma_f32_16x16x32_bf16 tiles D[slot][row] = sum_d q[slot][d] * k[row][d]
// (A operand q from registers, B operand straight from global: a lane's 8 consecutive d of one row is one 16-byte
// load), softmax over the slots = 4 registers + 2 cross-lane steps per row instead of K full wave reductions; the
// per-(row, slot) scalars go to a wave-private LDS table.  Phase 2: lanes own 4 channels (8-byte row accesses) and
// run the rank-1 updates with the scalars broadcast from LDS.  (The all-VALU kernels above spend ~66 shuffles and
// 6 two-byte loads per row: 404 / 709 us per call at B=32, N=4096, D=192 -- ~20x off the HBM bound.)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 sbf16x8;
typedef __attribute__((ext_vector_type(4))) float sf32x4;
union SPk8 { sbf16x8 v; uint4 u; bf16_t e[8]; };
constexpr int MROWS = 256;                        // rows per block (4 waves x 64)

__device__ __forceinline__ sbf16x8 slot_frag(const bf16_t* base, int slot, int K, int D, int ks, int g) {
    SPk8 f;
    f.u = make_uint4(0, 0, 0, 0);
    if (slot < K) f.u = *reinterpret_cast<const uint4*>(base + (int64_t)slot * D + ks * 32 + g * 8);
    return f.v;
}

// ---- rank-1 sums over the rows on the matrix pipe ----
// updates[slot][c] = sum_rows ae[row][slot] v[row][c]   (forward)      dq[slot][c] = sum_rows dl[row][slot] k[row][c]   (backward)
// are products with the ROW as the reduction index.  As VALU rank-1 updates they cost 80 wave instructions per row
// (64 FMA + the LDS broadcasts of the 16 scalars, 48 of 64 lanes active at D = 192): 17 us of issue per SIMD at
// B = 32, N = 4096 against 12.5 us of HBM time -- the forward launch measured 31.9 us, the backward 40.2.  Here:
//   * phase 1 runs "swapped" (A = the 16 rows of k / v, B = the slot matrix), so a lane ends with slot (lane & 15) of
//     rows 4g .. 4g+3: the softmax over the slots is a 16-lane DPP reduction, and the lane's values of two consecutive
//     16-row tiles ARE the A fragment of a 32-row reduction step, with reduction slot j of lane group g standing for row
//     16 (j >> 2) + 4 g + (j & 3) of the step;
//   * the other operand (v or k rows, channel on the lane, the same row order) is gathered from a wave-private LDS
//     image of the 32 rows by ds_read_b64_tr_b16; row pitch 2 D + 32 bytes = 8 * odd dwords: the 8 rows one 32-lane
//     half touches sit on 8 different 8-bank groups, conflict-free;
//   * 2 * D / 16 MFMAs per 32 rows replace 2560 VALU instructions.  The image is filled by 16-byte row-contiguous loads
//     (the 32 rows are one contiguous 64 D-byte run of HBM); no workgroup barrier until the final 4-wave combine, whose
//     buffer reuses the wave's own image.
typedef __attribute__((ext_vector_type(4))) short ss16x4;
typedef __attribute__((address_space(3))) ss16x4 lds_ss16x4;
union STrFrag { sbf16x8 v; ss16x4 t[2]; };

template <int KS> struct SlotImg {
    static constexpr int D = KS * 32;
    static constexpr int PITCH = 2 * D + 32;              // bytes per image row
    static constexpr int BYTES = 32 * PITCH;              // one wave's image: 32 rows (>= 16 x (D + 1) floats of the combine)
    static constexpr int CPR = D / 8;                     // 16-byte chunks per row
    static constexpr int NLD = 2 * KS;                    // chunks per lane
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// all-reduce over the 16 lanes of a DPP row: mirror (i <-> 15-i), half mirror (i <-> 7-i), quad xor 1, quad xor 2
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, dpp_f<0x140>(x)); x = fmaxf(x, dpp_f<0x141>(x)); x = fmaxf(x, dpp_f<0xB1>(x)); return fmaxf(x, dpp_f<0x4E>(x));
}
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_f<0x140>(x); x += dpp_f<0x141>(x); x += dpp_f<0xB1>(x); return x + dpp_f<0x4E>(x);
}

// ---- loads issued by hand (cdna_hip_programming.md 5.7 form (iii)) ----
// Left to the compiler, every group of loads sinks to its first use and the step becomes three exposed HBM round trips
// with 6-12 KB in flight per wave (25 us per launch); __builtin_amdgcn_sched_barrier does not help, the loads are
// sunk before the machine scheduler runs.  Here a step's loads are issued together, in the order they are needed, and
// each consumer waits with a counted s_waitcnt: vmcnt(number of HAND-ISSUED loads younger than the one needed).  The
// count ignores the stores (and, backward, nothing else reads through the compiler inside the step): loads return in
// order, so "at most Y operations outstanding" with Y younger loads can only hold once the awaited load is back --
// extra outstanding operations only make the wait longer.  The next step's loads are issued before this step's MFMA
// phase, into the registers phase 1 and the LDS copy have just released.
typedef __attribute__((ext_vector_type(4))) uint32_t su32x4;   // (a true vector type: HIP's uint4 struct is no asm register operand)
template <int OFF>
__device__ __forceinline__ void sload16(su32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void sload2(uint32_t& dst, const void* p) {
    asm volatile("global_load_ushort %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void sload4(uint32_t& dst, const void* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void swait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void spin(su32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void spin(uint32_t& v) { asm volatile("" : "+v"(v)); }

// the KS operand fragments of one 16-row tile: lane (row t16, d-group g) reads 16 bytes of every 64-byte k step
// HAND = false (row counts with a ragged last workgroup): the same loads through the compiler, which then also does
// the waiting.  The clamped per-chunk addresses raise the register pressure, and under pressure the compiler parks
// values in AGPRs -- a copy of a register whose hand-issued load is still in flight copies garbage (seen in the D = 256
// instance: v_accvgpr_write of the destination right after the global_load).  tests/test_asm_load_lint.py checks that no
// kernel with hand-issued loads contains such copies or scratch.
template <int KS, bool HAND, int... J>
__device__ __forceinline__ void slot_issue_frags(su32x4 (&f)[KS], const bf16_t* row, std::integer_sequence<int, J...>) {
    if constexpr (HAND) (sload16<J * 64>(f[J], row), ...);
    else ((f[J] = *reinterpret_cast<const su32x4*>(row + J * 32)), ...);
}
template <int KS, bool HAND, int... J>
__device__ __forceinline__ void slot_pin_frags(su32x4 (&f)[KS], std::integer_sequence<int, J...>) {
    if constexpr (HAND) (spin(f[J]), ...);
}
template <bool HAND, int N> __device__ __forceinline__ void slot_wait() { if constexpr (HAND) swait<N>(); }
__device__ __forceinline__ sbf16x8 as_frag(const su32x4& u) { return __builtin_bit_cast(sbf16x8, u); }

// rows n_first .. n_first+31 of a [N][D] matrix, 16 bytes per lane and load: chunk lane + 64 u of the 64 D-byte run.
// FULL (all 32 rows exist): hand-issued, one base address per 4 loads and immediate offsets; else compiler loads with
// every chunk's row clamped to N-1 (the weights of those rows are zero, the data only has to be finite).
template <int KS, bool FULL, int... U>
__device__ __forceinline__ void slot_issue_rows(su32x4 (&r)[2 * KS], const bf16_t* base, int n_first, int N, int lane,
                                                std::integer_sequence<int, U...>) {
    using I = SlotImg<KS>;
    if constexpr (FULL) {
        const char* p = reinterpret_cast<const char*>(base + (int64_t)n_first * I::D) + lane * 16;
        (sload16<(U & 3) * 1024>(r[U], p + (U >> 2) * 4096), ...);
    } else {
        auto one = [&](su32x4& dst, int u) __attribute__((always_inline)) {
            const int i = lane + 64 * u, row = i / I::CPR, ch = i - row * I::CPR;
            dst = *reinterpret_cast<const su32x4*>(base + (int64_t)min(n_first + row, N - 1) * I::D + ch * 8);
        };
        (one(r[U], U), ...);
    }
}
template <int KS, bool HAND, int... U>
__device__ __forceinline__ void slot_store_rows(char* img, su32x4 (&r)[2 * KS], int lane, std::integer_sequence<int, U...>) {
    using I = SlotImg<KS>;
    if constexpr (HAND) (spin(r[U]), ...);
#pragma unroll
    for (int u = 0; u < I::NLD; ++u) {
        const int i = lane + 64 * u, row = i / I::CPR, ch = i - row * I::CPR;
        *reinterpret_cast<su32x4*>(img + row * I::PITCH + ch * 16) = r[u];
    }
}
// acc[ct][r] += sum over the 32 rows of a[row][slot 4g + r] * img[row][16 ct + (lane & 15)]
template <int KS>
__device__ __forceinline__ void slot_rank_mfma(sf32x4 (&acc)[2 * KS], const char* img, sbf16x8 a, int lane) {
    using I = SlotImg<KS>;
    const int i = lane & 15, g = lane >> 4;
    const char* p0 = img + (4 * g + (i >> 2)) * I::PITCH + (i & 3) * 8;
#pragma unroll
    for (int ct = 0; ct < 2 * KS; ++ct) {
        STrFrag f;
        f.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ss16x4*)(p0 + ct * 32));
        f.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ss16x4*)(p0 + 16 * I::PITCH + ct * 32));
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, f.v, acc[ct], 0, 0, 0);
    }
}
// the 4 waves' [16 slots][D (+1)] accumulators meet in LDS (each wave writes into its own image), summed into partial
template <int KS, int EXTRA>
__device__ __forceinline__ void slot_combine(char* smem, const sf32x4 (&acc)[2 * KS], float extra, float* out, int K) {
    using I = SlotImg<KS>;
    constexpr int LD = I::D + EXTRA;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t16 = lane & 15, g = lane >> 4;
    float* mine = reinterpret_cast<float*>(smem + w * I::BYTES);
#pragma unroll
    for (int ct = 0; ct < 2 * KS; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(4 * g + r) * LD + 16 * ct + t16] = acc[ct][r];
    if (EXTRA && g == 0) mine[t16 * LD + I::D] = extra;
    __syncthreads();
    for (int e = threadIdx.x; e < K * LD; e += 256) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) t += reinterpret_cast<const float*>(smem + ww * I::BYTES)[e];
        out[e] = t;
    }
}

template <int KS, bool FULL>   // KS = D / 32; FULL: N is a multiple of the 256 rows of a workgroup
__global__ __launch_bounds__(256) void slot_fwd_mfma_kernel(const bf16_t* __restrict__ kt, const bf16_t* __restrict__ vt,
                                                            int64_t kv_bs, const bf16_t* __restrict__ q,
                                                            bf16_t* __restrict__ attn, int64_t attn_bs,
                                                            float* __restrict__ partial, int N, int K, float eps) {
    using I = SlotImg<KS>;
    constexpr int D = KS * 32;
    constexpr auto SK = std::make_integer_sequence<int, KS>{};
    constexpr auto SV = std::make_integer_sequence<int, 2 * KS>{};
    __shared__ __attribute__((aligned(16))) char smem[4 * I::BYTES];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t16 = lane & 15, g = lane >> 4;
    const int n0 = chunk * MROWS + w * 64;
    const bf16_t* kb = kt + (int64_t)b * kv_bs;
    const bf16_t* vb = vt + (int64_t)b * kv_bs;
    char* img = smem + w * I::BYTES;
    // DEEP (D <= 192): the NEXT step's k fragments are requested before this step's phase 1 starts (48 more registers in
    // flight), its v rows as soon as this step's are in LDS: HBM never idles behind the softmax arithmetic.
    constexpr bool DEEP = FULL && KS <= 6;
    su32x4 kf[2][2][KS], vr[2 * KS];                            // kf[step parity][tile][k step]
    auto issue_k = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
            slot_issue_frags<KS, FULL>(kf[DEEP ? step : 0][half], kb + (int64_t)min(n0 + step * 32 + half * 16 + t16, N - 1) * D + g * 8, SK);
    };
    auto issue_v = [&](int step) __attribute__((always_inline)) { slot_issue_rows<KS, FULL>(vr, vb, n0 + step * 32, N, lane, SV); };
    // the slot matrix first: it is older than everything below, so it is back when the first tile's wait returns (a
    // compiler-issued load here would make the compiler wait for ITS count, i.e. drain every prefetch)
    su32x4 qu[KS];
    slot_issue_frags<KS, FULL>(qu, q + ((int64_t)b * K + min(t16, K - 1)) * D + g * 8, SK);
    issue_k(0);
    issue_v(0);
    if (DEEP) issue_k(1);
    sbf16x8 qf[KS];
    sf32x4 U[2 * KS];
#pragma unroll
    for (int ct = 0; ct < 2 * KS; ++ct) U[ct] = (sf32x4){0.f, 0.f, 0.f, 0.f};
    float cs = 0.f;                                             // sum over this lane's rows of (attn + eps)[row][slot t16]
    const bool slot_ok = t16 < K;
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        // hand-issued loads younger than the one awaited (see the note on counting above)
        constexpr int AFTER_V = DEEP ? 2 * KS : 0;               // step 0 only: the next step's k fragments
        const int later = (DEEP && step == 0) ? AFTER_V : 0;
        SPk8 af;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nt = n0 + step * 32 + half * 16;          // first row of the 16-row tile
            if (half == 0) { if (later) slot_wait<FULL, 3 * KS + AFTER_V>(); else slot_wait<FULL, 3 * KS>(); }
            else { if (later) slot_wait<FULL, 2 * KS + AFTER_V>(); else slot_wait<FULL, 2 * KS>(); }
            su32x4 (&kt16)[KS] = kf[DEEP ? step : 0][half];
            slot_pin_frags<KS, FULL>(kt16, SK);
            if (step == 0 && half == 0) {
                slot_pin_frags<KS, FULL>(qu, SK);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[ks] = as_frag(t16 < K ? qu[ks] : (su32x4){0u, 0u, 0u, 0u});
            }
            sf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kt16[ks]), qf[ks], acc, 0, 0, 0);
            // acc[r] = logit[row nt + 4g + r][slot t16]: softmax over the 16 lanes of the DPP row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = slot_ok ? acc[r] : -INFINITY;
                const float m = row16_max(x);
                const float e = __expf(x - m);
                const float a = e * __builtin_amdgcn_rcpf(row16_sum(e));
                const int n = nt + 4 * g + r;
                const bool ok = slot_ok && n < N;
                if (ok) attn[(int64_t)b * attn_bs + (int64_t)n * K + t16] = f32_to_bf16(a);
                const float ae = ok ? a + eps : 0.f;
                cs += ae;
                af.e[4 * half + r] = f32_to_bf16(ae);
            }
        }
        if (later) slot_wait<FULL, AFTER_V>(); else slot_wait<FULL, 0>();
        slot_store_rows<KS, FULL>(img, vr, lane, SV);            // (LDS is in order per wave: the previous step's reads are done)
        if (step == 0) {
            if (!DEEP) issue_k(1);
            issue_v(1);
        }
        slot_rank_mfma<KS>(U, img, af.v, lane);
    }
    cs += __shfl_xor(cs, 16, 64);
    cs += __shfl_xor(cs, 32, 64);
    slot_combine<KS, 1>(smem, U, cs, partial + ((int64_t)b * nchunk + chunk) * K * (D + 1), K);
}

// Deferred backward of one corrector iteration: d(k_t), d(v_t) are left to slot_kv_grad_kernel (once per frame, for all
// iterations); this launch writes its rows of w = (attn + eps) / colsum and d(logits) to wl [B,N,32] bf16 (slots
// 0..15 | 16..31) and forms dq = sum_rows dlogits[row][slot] k[row][:] with the row-sum scheme above.
template <int KS, bool FULL, bool HAS_DA>
__global__ __launch_bounds__(256) void slot_bwd_defer_kernel(const bf16_t* __restrict__ kt, const bf16_t* __restrict__ vt,
                                                             int64_t kv_bs, const bf16_t* __restrict__ attn, int64_t attn_bs,
                                                             const float* __restrict__ colsum, const bf16_t* __restrict__ upd,
                                                             const bf16_t* __restrict__ dupd, const bf16_t* __restrict__ dattn,
                                                             bf16_t* __restrict__ wl, float* __restrict__ partial, int N, int K,
                                                             float eps) {
    using I = SlotImg<KS>;
    constexpr int D = KS * 32;
    constexpr int NA = HAS_DA ? 8 : 4;                          // scalar loads per tile (attn, d attn of the lane's 4 rows)
    constexpr auto SK = std::make_integer_sequence<int, KS>{};
    constexpr auto SV = std::make_integer_sequence<int, 2 * KS>{};
    __shared__ __attribute__((aligned(16))) char smem[4 * I::BYTES];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t16 = lane & 15, g = lane >> 4;
    const int n0 = chunk * MROWS + w * 64;
    const bf16_t* kb = kt + (int64_t)b * kv_bs;
    const bf16_t* vb = vt + (int64_t)b * kv_bs;
    char* img = smem + w * I::BYTES;
    const int tk = min(t16, K - 1);
    // The slot-side operands first (older than every row load: back when the first tile's wait returns; through the
    // compiler they would drain the prefetches): d(updates) and updates as fragments of slot t16, 1 / colsum[t16].
    su32x4 du[KS], uu[KS];
    uint32_t csu;
    slot_issue_frags<KS, FULL>(du, dupd + ((int64_t)b * K + tk) * D + g * 8, SK);
    slot_issue_frags<KS, FULL>(uu, upd + ((int64_t)b * K + tk) * D + g * 8, SK);
    if constexpr (FULL) sload4(csu, colsum + b * K + tk); else csu = __float_as_uint(colsum[b * K + tk]);
    su32x4 vf[2][KS], kr[2 * KS];
    uint32_t at[2][4], da[2][4];
    auto issue = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nt = n0 + step * 32 + half * 16;
            slot_issue_frags<KS, FULL>(vf[half], vb + (int64_t)min(nt + t16, N - 1) * D + g * 8, SK);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t o = (int64_t)b * attn_bs + (int64_t)min(nt + 4 * g + r, N - 1) * K + tk;
                if constexpr (FULL) {
                    sload2<0>(at[half][r], attn + o);
                    if (HAS_DA) sload2<0>(da[half][r], dattn + o);
                } else {
                    at[half][r] = attn[o];
                    if (HAS_DA) da[half][r] = dattn[o];
                }
            }
        }
        slot_issue_rows<KS, FULL>(kr, kb, n0 + step * 32, N, lane, SV);
    };
    issue(0);
    sbf16x8 df[KS];
    float my_r = 0.f, my_inv = 1.f;
    sf32x4 dQ[2 * KS];
#pragma unroll
    for (int ct = 0; ct < 2 * KS; ++ct) dQ[ct] = (sf32x4){0.f, 0.f, 0.f, 0.f};
    const bool slot_ok = t16 < K;
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        SPk8 lf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nt = n0 + step * 32 + half * 16;
            if (half == 0) slot_wait<FULL, 3 * KS + NA>(); else slot_wait<FULL, 2 * KS>();
            slot_pin_frags<KS, FULL>(vf[half], SK);
            if constexpr (FULL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { spin(at[half][r]); if (HAS_DA) spin(da[half][r]); }
            }
            if (step == 0 && half == 0) {
                // r[k] = d(updates)[k] . updates[k] = the diagonal of a 16 x 16 product of the two fragment sets: lane (k, g)
                // takes it from lane (k, k >> 2), register k & 3
                slot_pin_frags<KS, FULL>(du, SK);
                slot_pin_frags<KS, FULL>(uu, SK);
                if constexpr (FULL) spin(csu);
                sf32x4 pr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    df[ks] = as_frag(slot_ok ? du[ks] : (su32x4){0u, 0u, 0u, 0u});
                    pr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(uu[ks]), df[ks], pr, 0, 0, 0);
                }
                const int rr = t16 & 3;
                const float lo = (rr & 1) ? pr[1] : pr[0], hi = (rr & 1) ? pr[3] : pr[2];
                const float diag = (rr & 2) ? hi : lo;
                my_r = __shfl(diag, (t16 >> 2) * 16 + t16, 64);   // (valid in the source lanes g == t16 >> 2, which is what is read)
                my_inv = 1.f / __uint_as_float(csu);
            }
            sf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[half][ks]), df[ks], acc, 0, 0, 0);
            // acc[r] = dupd[slot t16] . v[row nt + 4g + r]: softmax backward over the 16 lanes of the DPP row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nt + 4 * g + r;
                const bool ok = slot_ok && n < N;
                float av = 0.f, dav = 0.f;
                if (ok) {
                    av = __uint_as_float(at[half][r] << 16);
                    dav = (acc[r] - my_r) * my_inv;
                    if (HAS_DA) dav += __uint_as_float(da[half][r] << 16);
                }
                const float dot = row16_sum(av * dav);
                const bf16_t wv = f32_to_bf16(ok ? (av + eps) * my_inv : 0.f);
                const bf16_t lv = f32_to_bf16(ok ? av * (dav - dot) : 0.f);
                if (n < N) {
                    bf16_t* o = wl + ((int64_t)b * N + n) * 32 + t16;
                    o[0] = wv;
                    o[16] = lv;
                }
                lf.e[4 * half + r] = lv;
            }
        }
        slot_wait<FULL, 0>();
        slot_store_rows<KS, FULL>(img, kr, lane, SV);
        if (step == 0) issue(1);
        slot_rank_mfma<KS>(dQ, img, lf.v, lane);
    }
    slot_combine<KS, 0>(smem, dQ, 0.f, partial + ((int64_t)b * nchunk + chunk) * K * D, K);
}

// Non-deferred backward (more than 4 iterations per frame, or the deferral switched off): d(k_t), d(v_t) are written (or
// accumulated) by this launch, dq included; the rank-1 updates of this variant stay on the VALU.
template <int KS>
__global__ __launch_bounds__(256) void slot_bwd_mfma_kernel(const bf16_t* __restrict__ kt, const bf16_t* __restrict__ vt,
                                                            int64_t kv_bs, const bf16_t* __restrict__ q,
                                                            const bf16_t* __restrict__ attn, int64_t attn_bs,
                                                            const float* __restrict__ colsum, const bf16_t* __restrict__ upd,
                                                            const bf16_t* __restrict__ dupd, const bf16_t* __restrict__ dattn,
                                                            bf16_t* __restrict__ dkt, bf16_t* __restrict__ dvt, int accumulate,
                                                            float* __restrict__ partial, int N, int K,
                                                            float eps) {
    constexpr int D = KS * 32;
    __shared__ __attribute__((aligned(16))) float sW[4][64][16];   // (attn + eps) / colsum per (row, slot)
    __shared__ __attribute__((aligned(16))) float sL[4][64][16];          // d logits per (row, slot)
    __shared__ float sr[16], scs[16];
    __shared__ float red[4][16][D];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t16 = lane & 15, g = lane >> 4;
    const int n0 = chunk * MROWS + w * 64;
    const bf16_t* kb = kt + (int64_t)b * kv_bs;
    const bf16_t* vb = vt + (int64_t)b * kv_bs;
    const bf16_t* qb = q + (int64_t)b * K * D;
    const bf16_t* dub = dupd + (int64_t)b * K * D;
    // r[k] = dupd[k,:] . upd[k,:]  (one wave per slot, round robin), colsum
    for (int k = w; k < K; k += 4) {
        float p = 0.f;
        for (int e = lane; e < D; e += 64) p += bf16_to_f32(dub[k * D + e]) * bf16_to_f32(upd[((int64_t)b * K + k) * D + e]);
        p = wave_sum(p);
        if (lane == 0) { sr[k] = p; scs[k] = colsum[b * K + k]; }
    }
    __syncthreads();
    sbf16x8 df[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) df[ks] = slot_frag(dub, t16, K, D, ks, g);
    // ---- phase 1: dw[slot][row] = dupd[slot] . v[row] on the matrix pipe; softmax backward per row ----
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
        const int n = n0 + tb * 16 + t16;
        const bool valid = n < N;
        const bf16_t* row = vb + (int64_t)min(n, N - 1) * D;
        sf32x4 acc = {0.f, 0.f, 0.f, 0.f};
        SPk8 vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks].u = *reinterpret_cast<const uint4*>(row + ks * 32 + g * 8);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[ks], vf[ks].v, acc, 0, 0, 0);
        float av[4], dav[4], dot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g + r;
            av[r] = 0.f; dav[r] = 0.f;
            if (valid && k < K) {
                const int64_t o = (int64_t)b * attn_bs + (int64_t)n * K + k;
                av[r] = bf16_to_f32(attn[o]);
                dav[r] = (acc[r] - sr[k]) / scs[k];
                if (dattn) dav[r] += bf16_to_f32(dattn[o]);
                dot += av[r] * dav[r];
            }
        }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        float4 wg, dl;
        float wv[4], lv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g + r;
            const bool ok = valid && k < K;
            wv[r] = ok ? (av[r] + eps) / scs[k] : 0.f;
            lv[r] = ok ? av[r] * (dav[r] - dot) : 0.f;
        }
        wg.x = wv[0]; wg.y = wv[1]; wg.z = wv[2]; wg.w = wv[3];
        dl.x = lv[0]; dl.y = lv[1]; dl.z = lv[2]; dl.w = lv[3];
        *reinterpret_cast<float4*>(&sW[w][tb * 16 + t16][4 * g]) = wg;
        *reinterpret_cast<float4*>(&sL[w][tb * 16 + t16][4 * g]) = dl;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- phase 2: lane = 4 channels.  dv[c] = sum_k w[k] dupd[k][c], dk[c] = sum_k dl[k] q[k][c], dQ[k][c] += dl[k] k[c]
    constexpr int NL = D / 4;
    const bool lact = lane < NL;
    float sq[16][4], sd[16][4], dQ[16][4];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = lact && k < K;
            sq[k][c] = ok ? bf16_to_f32(qb[k * D + lane * 4 + c]) : 0.f;
            sd[k][c] = ok ? bf16_to_f32(dub[k * D + lane * 4 + c]) : 0.f;
            dQ[k][c] = 0.f;
        }
    }
    for (int t0 = 0; t0 < 64; t0 += 4) {
        uint2 kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = min(n0 + t0 + u, N - 1);
            kr[u] = lact ? *reinterpret_cast<const uint2*>(kb + (int64_t)n * D + lane * 4) : make_uint2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + t0 + u;
            const float k0 = __uint_as_float(kr[u].x << 16), k1 = __uint_as_float(kr[u].x & 0xffff0000u);
            const float k2 = __uint_as_float(kr[u].y << 16), k3 = __uint_as_float(kr[u].y & 0xffff0000u);
            const float4* wr = reinterpret_cast<const float4*>(&sW[w][t0 + u][0]);
            const float4* lr = reinterpret_cast<const float4*>(&sL[w][t0 + u][0]);
            float ww[16], ll[16];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 y = lr[c4];
                ll[c4 * 4 + 0] = y.x; ll[c4 * 4 + 1] = y.y; ll[c4 * 4 + 2] = y.z; ll[c4 * 4 + 3] = y.w;
                const float4 x = wr[c4];
                ww[c4 * 4 + 0] = x.x; ww[c4 * 4 + 1] = x.y; ww[c4 * 4 + 2] = x.z; ww[c4 * 4 + 3] = x.w;
            }
            float dv[4] = {0.f, 0.f, 0.f, 0.f}, dk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { dv[c] = fmaf(ww[k], sd[k][c], dv[c]); dk[c] = fmaf(ll[k], sq[k][c], dk[c]); }
                dQ[k][0] = fmaf(ll[k], k0, dQ[k][0]); dQ[k][1] = fmaf(ll[k], k1, dQ[k][1]);
                dQ[k][2] = fmaf(ll[k], k2, dQ[k][2]); dQ[k][3] = fmaf(ll[k], k3, dQ[k][3]);
            }
            if (lact && n < N) {
                const int64_t ro = (int64_t)b * kv_bs + (int64_t)n * D + lane * 4;
                if (accumulate) {
                    const uint2 ok = *reinterpret_cast<const uint2*>(dkt + ro), ov = *reinterpret_cast<const uint2*>(dvt + ro);
                    dk[0] += __uint_as_float(ok.x << 16); dk[1] += __uint_as_float(ok.x & 0xffff0000u);
                    dk[2] += __uint_as_float(ok.y << 16); dk[3] += __uint_as_float(ok.y & 0xffff0000u);
                    dv[0] += __uint_as_float(ov.x << 16); dv[1] += __uint_as_float(ov.x & 0xffff0000u);
                    dv[2] += __uint_as_float(ov.y << 16); dv[3] += __uint_as_float(ov.y & 0xffff0000u);
                }
                uint2 o1, o2;
                o1.x = (uint32_t)f32_to_bf16(dk[0]) | ((uint32_t)f32_to_bf16(dk[1]) << 16);
                o1.y = (uint32_t)f32_to_bf16(dk[2]) | ((uint32_t)f32_to_bf16(dk[3]) << 16);
                o2.x = (uint32_t)f32_to_bf16(dv[0]) | ((uint32_t)f32_to_bf16(dv[1]) << 16);
                o2.y = (uint32_t)f32_to_bf16(dv[2]) | ((uint32_t)f32_to_bf16(dv[3]) << 16);
                *reinterpret_cast<uint2*>(dkt + ro) = o1;
                *reinterpret_cast<uint2*>(dvt + ro) = o2;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (lact) { red[w][k][lane * 4 + 0] = dQ[k][0]; red[w][k][lane * 4 + 1] = dQ[k][1]; red[w][k][lane * 4 + 2] = dQ[k][2]; red[w][k][lane * 4 + 3] = dQ[k][3]; }
    __syncthreads();
    float* out = partial + ((int64_t)b * nchunk + chunk) * K * D;
    for (int i = threadIdx.x; i < K * D; i += 256) {
        const int k = i / D, e = i - k * D;
        out[i] = red[0][k][e] + red[1][k][e] + red[2][k][e] + red[3][k][e];
    }
}


// ------------------------------------------------------------------------------------------------
// d(k_t), d(v_t) of ONE frame for all of its corrector iterations at once (steve.py:68-83 applies every iteration to the
// same k_t, v_t):
//     dk[n,:] = sum_i sum_k dlogits_i[n,k] q_i[k,:]        dv[n,:] = sum_i sum_k w_i[n,k] dupd_i[k,:]
// = two products [rows x (iterations x 16 slots)] . [(iterations x 16 slots) x D] on the matrix pipe.  The per-iteration
// backward launches only write their 64-byte (w, dlogits) rows; dk and dv (2 x 50 MB per frame at the BASELINE shape) are
// written once instead of three times written and twice re-read, and the 2 x 16 x D rank-1 updates per row leave the VALU.
// Block = 256 rows (4 waves x 4 tiles of 16); the stacked, transposed slot matrices [d][64 slots] sit in LDS as the MFMA
// A operand ("swapped" product: a lane ends with 4 consecutive channels of one row).
// ------------------------------------------------------------------------------------------------
struct KvGradArgs { const bf16_t* wl[4]; const bf16_t* q[4]; const bf16_t* du[4]; };
constexpr int SROW = 64 * 2 + 16;                 // bytes per channel row of the stacked slot image (64 slots + pad)
constexpr int KVSUB = 1;                          // row blocks of 256 per workgroup (B = 32, N = 4096: 512 workgroups, 2 per CU;
                                                  // with 2 a CU held one workgroup = 4 waves)

template <int KS>
__global__ __launch_bounds__(256) void slot_kv_grad_kernel(const KvGradArgs a, int iters, bf16_t* __restrict__ dkt,
                                                           bf16_t* __restrict__ dvt, int64_t kv_bs, int64_t kv_ld, int N, int K) {
    constexpr int D = KS * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][D][SROW]: Q^T stack, dU^T stack
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t16 = lane & 15, g = lane >> 4;
    // stacked transposed slot matrices: img[m][d][i*16 + k] = (m ? dupd_i : q_i)[b][k][d], zero for k >= K or i >= iters
    // (one 16-byte read of 8 channels of a slot row -> 8 two-byte LDS writes; a workgroup keeps the image for KVSUB
    // row blocks, so the transpose is amortised over 512 rows)
    // (the stacked slot is the fastest index over the lanes: a wave's 64 two-byte writes of one j land in one image row,
    // two lanes per dword; with the channel group fastest all 64 lanes hit one bank, 288 dwords apart)
#pragma unroll 4                                                      // (four slot-row loads in flight; rolled, 12 round trips in a row)
    for (int e = tid; e < 2 * 64 * (D / 8); e += 256) {
        const int sl = e & 63, rest = e >> 6, d8 = rest % (D / 8), m = rest / (D / 8);
        const int i = sl >> 4, k = sl & 15;
        SPk8 v;
        v.u = make_uint4(0, 0, 0, 0);
        if (i < iters && k < K) v.u = *reinterpret_cast<const uint4*>((m ? a.du[i] : a.q[i]) + ((int64_t)b * K + k) * D + d8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<bf16_t*>(smem + (m * D + d8 * 8 + j) * SROW + sl * 2) = v.e[j];
    }
    __syncthreads();
  for (int sub = 0; sub < KVSUB; ++sub) {
    const int n0 = (chunk * KVSUB + sub) * MROWS + w * 64;
    if (n0 >= N) break;
    // B operands (column = row n of the tile, k = 8 consecutive stacked slots): (w | dlogits) rows straight from HBM
    SPk8 fw[4][2], fl[4][2];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
        const int n = min(n0 + tb * 16 + t16, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int i = 2 * ks + (g >> 1);                      // iteration of this lane group's 8 slots
            fw[tb][ks].u = make_uint4(0, 0, 0, 0);
            fl[tb][ks].u = make_uint4(0, 0, 0, 0);
            if (i < iters) {
                const bf16_t* row = a.wl[i] + ((int64_t)b * N + n) * 32 + 8 * (g & 1);
                fw[tb][ks].u = *reinterpret_cast<const uint4*>(row);
                fl[tb][ks].u = *reinterpret_cast<const uint4*>(row + 16);
            }
        }
    }
    // Two 16-channel tiles per pass, their M rows interleaved so that a lane ends with EIGHT consecutive channels of its
    // row: M row 4 p + e of tile j is channel 32 dt + 8 p + 4 j + e, i.e. lane (row n, g) holds channels 32 dt + 8 g .. + 7
    // and a store instruction writes 16 rows x 64 contiguous bytes (8-byte stores in 32-byte pieces measured 47 us per
    // frame against 16 us of HBM time).
#pragma unroll 1
    for (int dt = 0; dt < D / 32; ++dt) {
        SPk8 aq[2][2], au[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = dt * 32 + 8 * (t16 >> 2) + 4 * j + (t16 & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                aq[j][ks].u = *reinterpret_cast<const uint4*>(smem + ch * SROW + (ks * 32 + g * 8) * 2);
                au[j][ks].u = *reinterpret_cast<const uint4*>(smem + (D + ch) * SROW + (ks * 32 + g * 8) * 2);
            }
        }
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            sf32x4 ck[2], cv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ck[j] = (sf32x4){0.f, 0.f, 0.f, 0.f};
                cv[j] = (sf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    ck[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[j][ks].v, fl[tb][ks].v, ck[j], 0, 0, 0);
                    cv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(au[j][ks].v, fw[tb][ks].v, cv[j], 0, 0, 0);
                }
            }
            const int n = n0 + tb * 16 + t16;                     // lane (row n, g): channels dt*32 + 8 g .. + 7
            if (n < N) {
                const int64_t ro = (int64_t)b * kv_bs + (int64_t)n * kv_ld + dt * 32 + 8 * g;
                uint4 o1, o2;
                o1.x = (uint32_t)f32_to_bf16(ck[0][0]) | ((uint32_t)f32_to_bf16(ck[0][1]) << 16);
                o1.y = (uint32_t)f32_to_bf16(ck[0][2]) | ((uint32_t)f32_to_bf16(ck[0][3]) << 16);
                o1.z = (uint32_t)f32_to_bf16(ck[1][0]) | ((uint32_t)f32_to_bf16(ck[1][1]) << 16);
                o1.w = (uint32_t)f32_to_bf16(ck[1][2]) | ((uint32_t)f32_to_bf16(ck[1][3]) << 16);
                o2.x = (uint32_t)f32_to_bf16(cv[0][0]) | ((uint32_t)f32_to_bf16(cv[0][1]) << 16);
                o2.y = (uint32_t)f32_to_bf16(cv[0][2]) | ((uint32_t)f32_to_bf16(cv[0][3]) << 16);
                o2.z = (uint32_t)f32_to_bf16(cv[1][0]) | ((uint32_t)f32_to_bf16(cv[1][1]) << 16);
                o2.w = (uint32_t)f32_to_bf16(cv[1][2]) | ((uint32_t)f32_to_bf16(cv[1][3]) << 16);
                *reinterpret_cast<uint4*>(dkt + ro) = o1;
                *reinterpret_cast<uint4*>(dvt + ro) = o2;
            }
        }
    }
  }
}

// FOCUS_SLOT_HAND=0: the compiler-issued variants (FULL = false) also where every row exists -- the same arithmetic with
// hipcc's own s_waitcnt in place of the hand-counted ones; read per call so that one test can compare the two bit for bit.
static bool slot_hand_loads() {
    const char* e = getenv("FOCUS_SLOT_HAND");
    return !(e && atoi(e) == 0);
}

inline bool slot_mfma_ok(const void* a, const void* b, const void* c, int64_t kv_bs, int K, int D, int dtype) {
    static const bool enabled = !(getenv("FOCUS_SLOT_MFMA") && atoi(getenv("FOCUS_SLOT_MFMA")) == 0);
    return enabled && dtype == FOCUS_BF16 && K <= 16 && (D == 64 || D == 128 || D == 192 || D == 256) && (kv_bs & 7) == 0 &&
           focus_aligned(a, 16) && focus_aligned(b, 16) && focus_aligned(c, 16);
}

inline int nchunks(int N) { return (N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }
inline int nchunks_mfma(int N) { return (N + MROWS - 1) / MROWS; }

}  // namespace

extern "C" size_t focus_slot_attn_workspace_bytes(int B, int N, int K, int D) {
    return (size_t)B * nchunks(N) * K * (D + 1) * sizeof(float);     // nchunks(N) >= nchunks_mfma(N)
}

#define SLOT_DISPATCH(KERNEL_CALL)                                                          \
    do {                                                                                    \
        const int dv = (D + 63) / 64;                                                       \
        const int kp = K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : 32;                         \
        bool done = false;                                                                  \
        SLOT_CASE(4, 1) SLOT_CASE(4, 2) SLOT_CASE(4, 3) SLOT_CASE(4, 4)                     \
        SLOT_CASE(8, 1) SLOT_CASE(8, 2) SLOT_CASE(8, 3) SLOT_CASE(8, 4)                     \
        SLOT_CASE(16, 1) SLOT_CASE(16, 2) SLOT_CASE(16, 3) SLOT_CASE(16, 4)                 \
        SLOT_CASE(32, 1) SLOT_CASE(32, 2) SLOT_CASE(32, 3) SLOT_CASE(32, 4)                 \
        if (!done) return FOCUS_ERR_SHAPE;                                                  \
    } while (0)

extern "C" int focus_slot_attn_fwd(const void* k_t, const void* v_t, int64_t kv_bs, const void* q, void* attn_vis,
                                   int64_t attn_bs, void* upd, float* colsum, void* partial, size_t partial_bytes,
                                   int B, int N, int K, int D, float eps, int dtype, void* stream) {
    if (!k_t || !v_t || !q || !attn_vis || !upd || !colsum || !partial) return FOCUS_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || K > 32 || D <= 0 || D > 256 || B > 65535) return FOCUS_ERR_SHAPE;
    if (partial_bytes < focus_slot_attn_workspace_bytes(B, N, K, D)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (slot_mfma_ok(k_t, v_t, q, kv_bs, K, D, dtype)) {
        dim3 gm(nchunks_mfma(N), B);
#define SFM_(KS, FULL) hipLaunchKernelGGL((slot_fwd_mfma_kernel<KS, FULL>), gm, dim3(256), 0, s, (const bf16_t*)k_t, (const bf16_t*)v_t, kv_bs, (const bf16_t*)q, (bf16_t*)attn_vis, attn_bs, (float*)partial, N, K, eps)
#define SFM(KS) do { if (N % MROWS == 0 && slot_hand_loads()) SFM_(KS, true); else SFM_(KS, false); } while (0)
        if (D == 64) SFM(2); else if (D == 128) SFM(4); else if (D == 192) SFM(6); else SFM(8);
#undef SFM
#undef SFM_
        FOCUS_CHECK_LAUNCH();
        hipLaunchKernelGGL((slot_fwd_finish<bf16_t>), dim3(K, B), dim3(64 * FIN_CL), 0, s, (const float*)partial, (bf16_t*)upd, colsum,
                           nchunks_mfma(N), K, D);
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    dim3 grid(nchunks(N), B);
    const size_t lds = ((size_t)K * D + 256) * sizeof(float);
#define SLOT_CASE(KP_, DV_)                                                                                   \
    if (!done && kp == KP_ && dv == DV_) {                                                                    \
        done = true;                                                                                          \
        if (dtype == FOCUS_BF16)                                                                              \
            hipLaunchKernelGGL((slot_fwd_kernel<bf16_t, KP_, DV_>), grid, dim3(256), lds, s, (const bf16_t*)k_t, \
                               (const bf16_t*)v_t, kv_bs, (const bf16_t*)q, (bf16_t*)attn_vis, attn_bs,       \
                               (float*)partial, N, K, D, eps);                                                \
        else                                                                                                  \
            hipLaunchKernelGGL((slot_fwd_kernel<float, KP_, DV_>), grid, dim3(256), lds, s, (const float*)k_t, \
                               (const float*)v_t, kv_bs, (const float*)q, (float*)attn_vis, attn_bs,          \
                               (float*)partial, N, K, D, eps);                                                \
    }
    SLOT_DISPATCH();
#undef SLOT_CASE
    FOCUS_CHECK_LAUNCH();
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((slot_fwd_finish<bf16_t>), dim3(K, B), dim3(64 * FIN_CL), 0, s, (const float*)partial, (bf16_t*)upd,
                           colsum, nchunks(N), K, D);
    else
        hipLaunchKernelGGL((slot_fwd_finish<float>), dim3(K, B), dim3(64 * FIN_CL), 0, s, (const float*)partial, (float*)upd,
                           colsum, nchunks(N), K, D);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_slot_attn_bwd(const void* k_t, const void* v_t, int64_t kv_bs, const void* q,
                                   const void* attn_vis, int64_t attn_bs, const float* colsum, const void* upd,
                                   const void* dupd, const void* dattn_vis, void* dk_t, void* dv_t, int accumulate,
                                   void* dq, void* partial, size_t partial_bytes, int B, int N, int K, int D,
                                   float eps, int dtype, void* wl, void* stream) {
    if (!k_t || !v_t || !q || !attn_vis || !colsum || !upd || !dupd || !dq || !partial) return FOCUS_ERR_NULL;
    if (!wl && (!dk_t || !dv_t)) return FOCUS_ERR_NULL;
    if (wl && !focus_slot_kv_grad_ok(K, D, dtype, 1)) return FOCUS_ERR_SHAPE;
    if (B <= 0 || N <= 0 || K <= 0 || K > 32 || D <= 0 || D > 256 || B > 65535) return FOCUS_ERR_SHAPE;
    if (partial_bytes < focus_slot_attn_workspace_bytes(B, N, K, D)) return FOCUS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (wl && !(slot_mfma_ok(k_t, v_t, q, kv_bs, K, D, dtype) && focus_aligned(wl, 16))) return FOCUS_ERR_ALIGN;
    if (slot_mfma_ok(k_t, v_t, q, kv_bs, K, D, dtype) && (wl || (focus_aligned(dk_t, 8) && focus_aligned(dv_t, 8)))) {
        dim3 gm(nchunks_mfma(N), B);
#define SBD(KS, FULL, DA) hipLaunchKernelGGL((slot_bwd_defer_kernel<KS, FULL, DA>), gm, dim3(256), 0, s, (const bf16_t*)k_t, (const bf16_t*)v_t, kv_bs, (const bf16_t*)attn_vis, attn_bs, colsum, (const bf16_t*)upd, (const bf16_t*)dupd, (const bf16_t*)dattn_vis, (bf16_t*)wl, (float*)partial, N, K, eps)
#define SBM(KS) do { \
        if (wl) { \
            if (N % MROWS == 0 && slot_hand_loads()) { if (dattn_vis) SBD(KS, true, true); else SBD(KS, true, false); } \
            else { if (dattn_vis) SBD(KS, false, true); else SBD(KS, false, false); } } \
        else hipLaunchKernelGGL((slot_bwd_mfma_kernel<KS>), gm, dim3(256), 0, s, (const bf16_t*)k_t, (const bf16_t*)v_t, kv_bs, (const bf16_t*)q, (const bf16_t*)attn_vis, attn_bs, colsum, (const bf16_t*)upd, (const bf16_t*)dupd, (const bf16_t*)dattn_vis, (bf16_t*)dk_t, (bf16_t*)dv_t, accumulate, (float*)partial, N, K, eps); } while (0)
        if (D == 64) SBM(2); else if (D == 128) SBM(4); else if (D == 192) SBM(6); else SBM(8);
#undef SBM
#undef SBD
        FOCUS_CHECK_LAUNCH();
        hipLaunchKernelGGL((slot_bwd_finish<bf16_t>), dim3(K, B), dim3(64 * FIN_CL), 0, s, (const float*)partial, (bf16_t*)dq,
                           nchunks_mfma(N), K, D);
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    dim3 grid(nchunks(N), B);
    const size_t lds = ((size_t)2 * K * D + 64 + 256) * sizeof(float);
#define SLOT_CASE(KP_, DV_)                                                                                    \
    if (!done && kp == KP_ && dv == DV_) {                                                                     \
        done = true;                                                                                           \
        if (dtype == FOCUS_BF16)                                                                               \
            hipLaunchKernelGGL((slot_bwd_kernel<bf16_t, KP_, DV_>), grid, dim3(256), lds, s, (const bf16_t*)k_t, \
                               (const bf16_t*)v_t, kv_bs, (const bf16_t*)q, (const bf16_t*)attn_vis, attn_bs,  \
                               colsum, (const bf16_t*)upd, (const bf16_t*)dupd, (const bf16_t*)dattn_vis,      \
                               (bf16_t*)dk_t, (bf16_t*)dv_t, accumulate, (float*)partial, N, K, D, eps);       \
        else                                                                                                   \
            hipLaunchKernelGGL((slot_bwd_kernel<float, KP_, DV_>), grid, dim3(256), lds, s, (const float*)k_t, \
                               (const float*)v_t, kv_bs, (const float*)q, (const float*)attn_vis, attn_bs,     \
                               colsum, (const float*)upd, (const float*)dupd, (const float*)dattn_vis,         \
                               (float*)dk_t, (float*)dv_t, accumulate, (float*)partial, N, K, D, eps);         \
    }
    SLOT_DISPATCH();
#undef SLOT_CASE
    FOCUS_CHECK_LAUNCH();
    if (dtype == FOCUS_BF16)
        hipLaunchKernelGGL((slot_bwd_finish<bf16_t>), dim3(K, B), dim3(64 * FIN_CL), 0, s, (const float*)partial, (bf16_t*)dq,
                           nchunks(N), K, D);
    else
        hipLaunchKernelGGL((slot_bwd_finish<float>), dim3(K, B), dim3(64 * FIN_CL), 0, s, (const float*)partial, (float*)dq,
                           nchunks(N), K, D);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_slot_kv_grad_ok(int K, int D, int dtype, int iters) {
    static const bool enabled = !(getenv("FOCUS_SLOT_KV_DEFER") && atoi(getenv("FOCUS_SLOT_KV_DEFER")) == 0);
    return enabled && dtype == FOCUS_BF16 && K >= 1 && K <= 16 && (D == 64 || D == 128 || D == 192 || D == 256) && iters >= 1 &&
           iters <= 4;
}

extern "C" int focus_slot_kv_grad(const void* wl0, const void* wl1, const void* wl2, const void* wl3, const void* q0,
                                  const void* q1, const void* q2, const void* q3, const void* du0, const void* du1,
                                  const void* du2, const void* du3, int iters, void* dk_t, void* dv_t, int64_t kv_bs,
                                  int64_t kv_ld, int B, int N, int K, int D, int dtype, void* stream) {
    if (!dk_t || !dv_t) return FOCUS_ERR_NULL;
    if (B <= 0 || N <= 0 || B > 65535 || !focus_slot_kv_grad_ok(K, D, dtype, iters) || (kv_bs & 7) || kv_ld < D || (kv_ld & 7))
        return FOCUS_ERR_SHAPE;
    KvGradArgs a;
    const void* wl[4] = {wl0, wl1, wl2, wl3};
    const void* q[4] = {q0, q1, q2, q3};
    const void* du[4] = {du0, du1, du2, du3};
    for (int i = 0; i < 4; ++i) {
        if (i < iters && (!wl[i] || !q[i] || !du[i])) return FOCUS_ERR_NULL;
        if (i < iters && !focus_aligned(wl[i], 16)) return FOCUS_ERR_ALIGN;
        a.wl[i] = (const bf16_t*)wl[i]; a.q[i] = (const bf16_t*)q[i]; a.du[i] = (const bf16_t*)du[i];
    }
    if (!focus_aligned(dk_t, 16) || !focus_aligned(dv_t, 16) || (kv_bs & 7)) return FOCUS_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((nchunks_mfma(N) + KVSUB - 1) / KVSUB, B);
    const size_t lds = (size_t)2 * D * SROW;
#define SKV(KS) do { \
        static bool once = (hipFuncSetAttribute((const void*)slot_kv_grad_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
        (void)once; \
        hipLaunchKernelGGL((slot_kv_grad_kernel<KS>), grid, dim3(256), lds, s, a, iters, (bf16_t*)dk_t, (bf16_t*)dv_t, kv_bs, kv_ld, N, K); } while (0)
    if (D == 64) SKV(2); else if (D == 128) SKV(4); else if (D == 192) SKV(6); else SKV(8);
#undef SKV
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
