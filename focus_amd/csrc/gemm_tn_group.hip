// gemm_tn_group.hip -- the weight (and bias) gradients of SEVERAL nn.Linear layers in ONE persistent launch:
//     dW_p[N_p, K_p] = dY_p[M_p, N_p]^T . X_p[M_p, K_p],   db_p[N_p] = sum_m dY_p[m, :]        p = 0 .. n-1
// (autograd of attention.py:506,536,555 and common.py:26-34 inside one Motionformer / ORViT block).
//
// Why: one Linear's weight gradient at the bench shape is 18-72 output tiles of 256 x 128 with a 12552-row reduction.
// Alone it cannot fill 256 CUs without splitting that reduction 4-14 ways: every split writes an fp32 partial tile to a
// slab and a second launch sums the slabs (profiles/r03_base: 768 x 768 x 12552 at 421 TF/s, 67 reduce launches per step).
// The five Linears of a block together are 234 tiles: ONE unit per tile covers the machine with NO split, no slabs and no
// reduce launch; every unit streams the whole reduction of its tile and stores its fp32 tile straight to dW.
//
// Same workgroup as gemm_tn_ws_kernel<256, 128> (gemm_mfma_tn_ws.hip: 8 consumer + 4 loader waves, 3-stage LDS ring,
// transposed LDS reads for the reduction-major operands); the problem table (up to 8 Linears) travels as a kernel argument.
// The unit list (built once per shape signature by focus_linear_wgrad_group_plan, kept on the device by the caller) walks
// every problem's tile grid in blocks of 4 x 8 tiles and every XCD owns a contiguous band of the list (~one block): the
// ~29 units that share an L2 read 4 dY panels and 8 X panels between them.  (First version: tile_j fastest over whole
// rows -- PMC FETCH 720 MB per launch against ~250 MB of operands: every XCD streamed most X panels.)  Bias gradients ride on the matrix pipe as ones^T . dY (unit
// tile_j takes the K-steps kt = tile_j mod tiles_j) and are ADDED to db with fp32 atomics (a few thousand per launch);
// db is zeroed by the caller's memset.
#include "focus_common.h"
#include "gemm_internal.h"
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BKM = 64;                 // reduction rows per K-step
constexpr int SUB = BKM * 256;          // one [64 rows][128 columns] bf16 sub-tile: 16 KiB, 256-B rows
constexpr int NSTAGE = 3, NLOAD = 4;
constexpr int BI = 256, BJ = 128;

union Frag { bf16x8 v; s16x4 t[2]; uint32_t u[4]; };

__device__ __forceinline__ int swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }
__device__ __forceinline__ int toff(int row, int col) {
    return row * 256 + ((((col >> 3)) ^ swz(row)) << 4) + (col & 4) * 2;
}
__device__ __forceinline__ bf16x8 col_frag16(const char* tile, int r0, int c0, int lane) {
    const int i = lane & 15;
    const int row = r0 + (i >> 2), col = c0 + 4 * (i & 3);
    Frag f;
    f.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + toff(row, col)));
    f.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + toff(row + 4, col)));
    return f.v;
}

// problem record (kernel argument)
struct Prob {
    const bf16_t* P;      // dY [Mred, I], row stride ldp
    const bf16_t* Q;      // X  [Mred, J], row stride ldq
    float* C;             // dW [I, J] dense
    float* db;            // [I] or NULL
    int64_t ldp, ldq;
    int32_t I, J, Mred, tiles_i, tiles_j, pad_;
};
constexpr int MAXP = 8;
struct GroupArgs { Prob p[MAXP]; int32_t nprob, nunits; };
struct UnitRec { int16_t prob, ti, tj, pad_; };     // 8 bytes

__global__ __launch_bounds__(64 * (BI * BJ / 4096 + NLOAD)) void gemm_tn_group_kernel(const GroupArgs ga, const UnitRec* __restrict__ units) {
    constexpr int NCONS = BI * BJ / 4096;
    constexpr int WJ = BJ / 64;
    constexpr int NSP = BI / 128, NSQ = BJ / 128, NSUBT = NSP + NSQ;
    constexpr int STAGE = NSUBT * SUB;
    constexpr int PIECES = NSUBT * 16 / NLOAD;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-banded schedule over the logical unit list (workgroup ids are dealt round-robin over the 8 XCDs)
    const int nunits = ga.nunits;
    const int G = gridDim.x, xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q8 = nunits >> 3, r8 = nunits & 7;
    const int band0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int band_n = q8 + (xcd < r8 ? 1 : 0);
    const int my_units = jwg < band_n ? (band_n - jwg + gx - 1) / gx : 0;
    if (my_units == 0) return;

    struct Unit { const bf16_t* P; const bf16_t* Q; float* C; float* db; int64_t ldp, ldq; int I, J, Mred, i0, j0, nk, valid_last, tj, tiles_j; };
    auto unit_of = [&](int i) __attribute__((always_inline)) {
        const UnitRec ur = units[band0 + jwg + i * gx];
        const int pi = __builtin_amdgcn_readfirstlane((int)ur.prob);
        Unit t;
        // (selected with a chain of compares on the argument registers: the table lives in SGPRs / constant memory)
        Prob p = ga.p[0];
#pragma unroll
        for (int k = 1; k < MAXP; ++k)
            if (pi == k) p = ga.p[k];
        t.P = p.P; t.Q = p.Q; t.C = p.C; t.db = p.db; t.ldp = p.ldp; t.ldq = p.ldq; t.I = p.I; t.J = p.J; t.Mred = p.Mred;
        t.i0 = __builtin_amdgcn_readfirstlane((int)ur.ti) * BI;
        t.tj = __builtin_amdgcn_readfirstlane((int)ur.tj);
        t.j0 = t.tj * BJ;
        t.tiles_j = p.tiles_j;
        t.nk = (p.Mred + BKM - 1) / BKM;
        t.valid_last = p.Mred - (t.nk - 1) * BKM;
        return t;
    };

    if (w >= NCONS) {
        // =============================== loader waves ===============================
        const int L = w - NCONS;
        const int cpos = lane & 15, rin = lane >> 4;
        Unit cur = unit_of(0);
        int iu = 0, ikt = 0;
        auto issue = [&](int st) __attribute__((always_inline)) {
            char* sp = smem + st * STAGE;
            const int mb = ikt * BKM;
#pragma unroll
            for (int g = 0; g < PIECES; ++g) {
                const int sub = g >> 2, pi = g * NLOAD + L, row = (pi & 15) * 4 + rin;
                const int m = min(mb + row, cur.Mred - 1);
                const int ch = cpos ^ swz(row);
                const bf16_t* src;
                if (sub < NSP) src = cur.P + (int64_t)m * cur.ldp + min(cur.i0 + sub * 128 + ch * 8, cur.I - 8);
                else src = cur.Q + (int64_t)m * cur.ldq + min(cur.j0 + (sub - NSP) * 128 + ch * 8, cur.J - 8);
                __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(sp + pi * 1024), 16, 0, 0);
            }
            if (++ikt == cur.nk) { ikt = 0; if (++iu < my_units) cur = unit_of(iu); }
        };
        int total = 0;
        for (int i = 0; i < my_units; ++i) total += unit_of(i).nk;
        issue(0);
        if (total > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                      // step 0 is in LDS
        int st2 = 2;
        for (int t = 0; t < total; ++t) {
            if (t + 2 < total) {
                issue(st2);
                st2 = st2 == 2 ? 0 : st2 + 1;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");     // step t+1 landed, t+2 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                  // end of K-step t
        }
        return;
    }

    // =============================== consumer waves ===============================
    const int wi = w / WJ, wj = w % WJ;
    const int fr = lane & 15, fq = lane >> 4;
    const int poff = (wi >> 1) * SUB, pcol = (wi & 1) * 64;
    const int qoff = (NSP + (wj >> 1)) * SUB, qcol = (wj & 1) * 64;
    f32x4 acc[4][4];
    constexpr int NA = 4 / WJ;
    f32x4 accb[NA];
    Frag ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3F803F80u;
    auto compute = [&](const char* stage, auto masked, int valid, bool cs) __attribute__((always_inline)) {
        const char* sp = stage + poff;
        const char* sq = stage + qoff;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fp[4], fqv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fp[a] = col_frag16(sp, ks * 32 + 8 * fq, pcol + a * 16, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) fqv[b] = col_frag16(sq, ks * 32 + 8 * fq, qcol + b * 16, lane);
            if constexpr (decltype(masked)::value) {
                const int left = valid - (ks * 32 + 8 * fq);            // elements e < left are real
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    Frag f; f.v = fp[a];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const uint32_t keep = (2 * e2 < left ? 0x0000ffffu : 0u) | (2 * e2 + 1 < left ? 0xffff0000u : 0u);
                        f.u[e2] &= keep;
                    }
                    fp[a] = f.v;
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqv[b], fp[a], acc[a][b], 0, 0, 0);
            if (cs) {                                   // wave-uniform; touches accb only (acc stays branch-free)
#pragma unroll
                for (int n = 0; n < NA; ++n) {
                    bf16x8 sel = fp[n];
#pragma unroll
                    for (int j = 1; j < WJ; ++j)
                        if (wj == j) sel = fp[j * NA + n];
                    accb[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, sel, accb[n], 0, 0, 0);
                }
            }
        }
    };

    __builtin_amdgcn_s_barrier();                          // step 0 is in LDS
    int st = 0;
    for (int cu = 0; cu < my_units; ++cu) {
        const Unit cur = unit_of(cu);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool want_b = cur.db != nullptr;
        int cs_in = cur.tj;                                // steps until this unit's next column-sum K-step
#pragma unroll
        for (int n = 0; n < NA; ++n) accb[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt + 1 < cur.nk; ++kt) {
            compute(smem + st * STAGE, std::false_type{}, BKM, want_b && cs_in == 0);
            cs_in = cs_in == 0 ? cur.tiles_j - 1 : cs_in - 1;
            st = st == 2 ? 0 : st + 1;
            __builtin_amdgcn_s_barrier();                  // end of this K-step
        }
        compute(smem + st * STAGE, std::true_type{}, cur.valid_last, want_b && cs_in == 0);
        st = st == 2 ? 0 : st + 1;
        __builtin_amdgcn_s_barrier();
        // acc[a][b][r4] = D[j = j0 + wj*64 + b*16 + fq*4 + r4][i = i0 + wi*64 + a*16 + fr]: straight from the registers
        if (want_b && fq == 0) {
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int gi = cur.i0 + wi * 64 + (wj * NA + n) * 16 + fr;
                if (gi < cur.I) atomicAdd(cur.db + gi, accb[n][0]);
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int gi = cur.i0 + wi * 64 + a * 16 + fr;
            if (gi >= cur.I) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int gj = cur.j0 + wj * 64 + b * 16 + fq * 4;
                if (gj < cur.J)      // J % 8 == 0: the 4 columns are in range together
                    *reinterpret_cast<float4*>(cur.C + (int64_t)gi * cur.J + gj) =
                        make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
            }
        }
    }
}

}  // namespace

// Units of the grouped launch for problems (N_p, K_p, M_p): n_units = sum of ceil(N/256) * ceil(K/128).
extern "C" int focus_linear_wgrad_group_units(const focus_wgrad_item* items, int n_items) {
    if (!items || n_items <= 0) return 0;
    int64_t t = 0;
    for (int p = 0; p < n_items; ++p) t += (int64_t)((items[p].N + BI - 1) / BI) * ((items[p].K + BJ - 1) / BJ);
    return t > 0x7fffffff ? -1 : (int)t;
}

// The unit list of a grouped launch (HOST memory, focus_linear_wgrad_group_units entries of 8 bytes): depends on the
// (N, K) of the problems only.  The caller copies it to the device once per shape signature and passes it to every launch.
extern "C" int focus_linear_wgrad_group_plan(const focus_wgrad_item* items, int n_items, void* host_units, size_t bytes) {
    if (!items || !host_units) return FOCUS_ERR_NULL;
    if (n_items <= 0 || n_items > MAXP) return FOCUS_ERR_SHAPE;
    const int total = focus_linear_wgrad_group_units(items, n_items);
    if (total <= 0 || bytes < (size_t)total * sizeof(UnitRec)) return FOCUS_ERR_WORKSPACE;
    UnitRec* u = static_cast<UnitRec*>(host_units);
    constexpr int BA = 4, BB = 8;               // tiles per block: 4 dY panels (256 columns) x 8 X panels (128 columns)
    int n = 0;
    for (int p = 0; p < n_items; ++p) {
        const int ti_n = (items[p].N + BI - 1) / BI, tj_n = (items[p].K + BJ - 1) / BJ;
        for (int bi = 0; bi < ti_n; bi += BA)
            for (int bj = 0; bj < tj_n; bj += BB)
                for (int ti = bi; ti < std::min(bi + BA, ti_n); ++ti)
                    for (int tj = bj; tj < std::min(bj + BB, tj_n); ++tj)
                        u[n++] = UnitRec{(int16_t)p, (int16_t)ti, (int16_t)tj, 0};
    }
    return n == total ? FOCUS_OK : FOCUS_ERR_SHAPE;
}

// `items` is a HOST array of at most 8 problems (longest reduction first gives the best balance); every db must have been
// zeroed by the caller (the bias gradients are accumulated with atomics); dW is written densely ([N, K], row stride K).
// dev_units: DEVICE copy of the list focus_linear_wgrad_group_plan made for the same (N, K) sequence.
extern "C" int focus_linear_wgrad_group(const focus_wgrad_item* items, int n_items, const void* dev_units, void* stream) {
    if (!items || !dev_units) return FOCUS_ERR_NULL;
    if (n_items <= 0) return FOCUS_OK;
    if (n_items > MAXP) return FOCUS_ERR_SHAPE;
    GroupArgs ga = {};
    int units = 0;
    for (int p = 0; p < n_items; ++p) {
        const focus_wgrad_item& it = items[p];
        if (!it.dy || !it.x || !it.dw) return FOCUS_ERR_NULL;
        if (it.M <= 0 || it.N < 8 || it.K < 8 || (it.N & 7) || (it.K & 7) || (it.ld_dy & 7) || (it.ld_x & 7)) return FOCUS_ERR_SHAPE;
        if (!focus_aligned(it.dy, 16) || !focus_aligned(it.x, 16) || !focus_aligned(it.dw, 16)) return FOCUS_ERR_ALIGN;
        Prob& q = ga.p[p];
        q.P = static_cast<const bf16_t*>(it.dy); q.Q = static_cast<const bf16_t*>(it.x);
        q.C = it.dw; q.db = it.db; q.ldp = it.ld_dy; q.ldq = it.ld_x;
        q.I = it.N; q.J = it.K; q.Mred = it.M;
        q.tiles_i = (it.N + BI - 1) / BI; q.tiles_j = (it.K + BJ - 1) / BJ; q.pad_ = 0;
        if (q.tiles_i > 32767 || q.tiles_j > 32767) return FOCUS_ERR_SHAPE;
        units += q.tiles_i * q.tiles_j;
    }
    ga.nprob = n_items;
    ga.nunits = units;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)NSTAGE * (BI / 128 + BJ / 128) * SUB;
    auto k = gemm_tn_group_kernel;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(k, dim3(std::min(units, 256)), dim3(64 * (BI * BJ / 4096 + NLOAD)), lds, s, ga,
                       static_cast<const UnitRec*>(dev_units));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
