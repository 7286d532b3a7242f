// slot_tail.hip -- the per-iteration "tail" of the STEVE slot update (steve.py:72-75, 85-93; STEVE/utils.py:107-118) as ONE
// launch per direction instead of eight (forward) / twelve (backward):
//     GRU cell:  gi = u W_ih^T + b_ih,  gh = h W_hh^T + b_hh,  r, z = sigmoid(..), n = tanh(gi_n + r gh_n),  h' = (1-z) n + z h
//     residual MLP (all but the last iteration):  s = h' + W2 relu(W1 LN_mlp(h') + b1) + b2
//     next iteration's query:  q = Wq LN_slots(s)
// on the B x K slot rows (352 at the BASELINE shape).  The chain is latency bound: every stage is a [rows x 192..768] product
// that cannot start before the previous one ends, and as separate launches each pays a kernel boundary (~1.5 us) plus a 4-6 us
// kernel whose only real work is streaming its weight matrix.  Here a workgroup owns 16 rows for the whole chain: activations
// stay in LDS, the weights (1.03 MB per workgroup) stream from L2 straight into MFMA B fragments (a lane's 8 consecutive k of
// one weight row = one 16-byte load), v_mfma_f32_16x16x32_bf16 with M = the 16 rows.  22 workgroups at B = 32, K = 11.
// Everything a backward needs is written as it is produced (bf16 storage rounds exactly where the unfused kernels round).
#include "focus_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int ROWS = 16;

__device__ __forceinline__ float sigm_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- the weight stream ------------------------------------------------------------------------------------------
// Every product of the chain is out[16][NOUT] = act[16][KIN] . W[NOUT][KIN]^T with the bf16 weights streamed from L2.  Wave w
// owns the 16-column output blocks nb = w, w+4, ... of every product; its share of ALL the weight matrices of the launch is
// one sequence of UNITS = [16 weight rows][192 k] (6 KiB: 6 LDS-DMA pieces of whole 128-byte row segments), and the wave
// streams that sequence through a private 4-slot LDS ring, 3 units ahead, ACROSS the stage boundaries (the weights do not
// depend on the activations: while the workgroup sits in the gate math or a LayerNorm, the next product's first blocks
// are already landing).  B fragments are read from the ring with ds_read_b128.
// (v1 loaded B fragments straight from global memory, 16 rows x 64 B per wave instruction: that access shape is served at
// ~16 B/clk/CU by the texture addresser -- 1 MB of weights per workgroup = 32-45 us per launch, as slow as the eight
// launches it replaced.  Row-contiguous DMA pieces are served at 2-4x that.)
constexpr int NSLOT = 4, UNIT_BYTES = 16 * 192 * 2, PIECES_PER_UNIT = UNIT_BYTES / 1024;

struct Seg { const bf16_t* W; int kin, nt, nkc; };           // nt: blocks of this wave, nkc: 192-wide k chunks per block

struct WStream {
    // issue-side iterator (the segments are read off the launch arguments: scalar registers)
    int is, it, ikc, issued, total;
    uint32_t ring;                                              // LDS byte address of this wave's ring
};

// segment i of the stream: [W_ih][W_hh] (GRU), [W1][W2] (MLP), [Wq]; a stage that is switched off has nt = 0
__device__ __forceinline__ Seg seg_at(const focus_slot_tail_args& a, int i) {
    const int D = a.D, H = a.H;
    Seg r = Seg{static_cast<const bf16_t*>(a.w_ih), D, a.do_gru ? 3 * D / 64 : 0, 1};
    if (i == 1) r = Seg{static_cast<const bf16_t*>(a.w_hh), D, a.do_gru ? 3 * D / 64 : 0, 1};
    if (i == 2) r = Seg{static_cast<const bf16_t*>(a.w1), D, a.do_mlp ? H / 64 : 0, 1};
    if (i == 3) r = Seg{static_cast<const bf16_t*>(a.w2), H, a.do_mlp ? D / 64 : 0, H / 192};
    if (i == 4) r = Seg{static_cast<const bf16_t*>(a.wq), D, a.do_q ? D / 64 : 0, 1};
    return r;
}

__device__ __forceinline__ Seg seg_at(const focus_slot_tail_bwd_args& a, int i);     // the backward's stream (below)

// DMA of the next unit of the stream into slot (issued % NSLOT); lane -> 16-byte chunk c = 64 p + lane of the [16][24] chunk
// grid; the chunk a row keeps at position j of an 8-chunk group is chunk j ^ ((row >> 1) & 7) (conflict-free fragment reads)
template <typename Args>
__device__ __forceinline__ void ws_issue(const Args& a, WStream& ws, int w, int lane) {
    if (ws.issued >= ws.total) return;
    const Seg sg = seg_at(a, ws.is);
    const int nb = w + 4 * ws.it;
    const char* base = reinterpret_cast<const char*>(sg.W) + ((int64_t)nb * 16 * sg.kin + ws.ikc * 192) * 2;
    const uint32_t slot = ws.ring + (ws.issued % NSLOT) * UNIT_BYTES;
#pragma unroll
    for (int p = 0; p < PIECES_PER_UNIT; ++p) {
        const int c = 64 * p + lane, row = c / 24, pos = c - row * 24;
        const int col = (pos & ~7) | ((pos & 7) ^ ((row >> 1) & 7));
        glds16(base + (int64_t)row * sg.kin * 2 + col * 16, __builtin_amdgcn_readfirstlane(slot + p * 1024));
    }
    ++ws.issued;
    if (++ws.ikc == sg.nkc) {
        ws.ikc = 0;
        if (++ws.it == sg.nt) {
            ws.it = 0;
            ++ws.is;
            while (ws.is < 5 && seg_at(a, ws.is).nt == 0) ++ws.is;
        }
    }
}

// wait until the unit `u` of this wave's stream has landed: everything issued after it may still be in flight
__device__ __forceinline__ void ws_wait(const WStream& ws, int u) {
    const int younger = ws.issued - u - 1;                      // units issued after u: 0 .. NSLOT-1
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one product: consumes this wave's next nt * nkc units; epi(column, first row, acc) per finished block
template <typename Args, typename Epi>
__device__ __forceinline__ void gemm16(const Args& a, WStream& ws, int& ucons, const char* ring_ptr, const bf16_t* sAct,
                                       int pitch, int nt, int nkc, int w, int lane, Epi epi) {
    const int row = lane & 15, kq = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            // the slot of the unit consumed last (its fragments sit in registers that the MFMAs have read) takes the unit NSLOT-1 ahead
            ws_issue(a, ws, w, lane);
            ws_wait(ws, ucons);
            const char* slot = ring_ptr + (ucons % NSLOT) * UNIT_BYTES;
            bf16x8 bfr[6], af[6];
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const int ch = ks * 4 + kq;
                bfr[ks] = *reinterpret_cast<const bf16x8*>(slot + row * 384 + (((ch & ~7) | ((ch & 7) ^ ((row >> 1) & 7))) << 4));
                af[ks] = *reinterpret_cast<const bf16x8*>(sAct + row * pitch + kc * 192 + ks * 32 + kq * 8);
            }
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bfr[ks], acc, 0, 0, 0);
            // the reads above are complete once the MFMAs have their operands: make that explicit before the slot is refilled
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++ucons;
        }
        epi((w + 4 * t) * 16 + row, kq * 4, acc);                // acc[r] = out[row kq*4 + r][column nb*16 + (lane & 15)]
    }
}

// LayerNorm of the 16 rows of an LDS tile (16 threads per row): writes the normalised bf16 rows to sOut and to gOut, the
// statistics to gMean / gRstd
template <int D>
__device__ __forceinline__ void ln16(const bf16_t* sIn, bf16_t* sOut, int pitch, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float eps, bf16_t* gOut, float* gMean, float* gRstd, int r0,
                                     int R, int tid) {
    constexpr int PER = D / 16;
    const int row = tid >> 4, l = tid & 15;
    float x[PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) { x[j] = bf16_to_f32(sIn[row * pitch + l + 16 * j]); s += x[j]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.f / D);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) { const float dlt = x[j] - mean; v += dlt * dlt; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = rsqrtf(v * (1.f / D) + eps);
    const bool live = r0 + row < R;
    if (l == 0 && live) { gMean[r0 + row] = mean; gRstd[r0 + row] = rstd; }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = l + 16 * j;
        const bf16_t y = f32_to_bf16((x[j] - mean) * rstd * gamma[c] + beta[c]);
        sOut[row * pitch + c] = y;
        if (live) gOut[(int64_t)(r0 + row) * D + c] = y;
    }
}

template <int D, int H>
__global__ __launch_bounds__(256) void slot_tail_fwd_kernel(const focus_slot_tail_args a) {
    constexpr int PD = D + 8, PH = H + 8, G3 = 3 * D;          // LDS pitches (elements): +16 bytes against bank conflicts
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sU = reinterpret_cast<bf16_t*>(smem);               // [16][PD]  updates; later the slots after the MLP
    bf16_t* sHp = sU + ROWS * PD;                               // [16][PD]  GRU state h (Q-only mode: the input slots)
    bf16_t* sHn = sHp + ROWS * PD;                              // [16][PD]  h' after the GRU
    bf16_t* sY = sHn + ROWS * PD;                               // [16][PD]  LayerNorm outputs
    bf16_t* sG = sY + ROWS * PD;                                // [2][16][3D] gate pre-activations | [16][PH] MLP hidden
    char* sRing = reinterpret_cast<char*>(sG + 2 * ROWS * G3);  // [4 waves][NSLOT][6 KiB] weight ring
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * ROWS, R = a.R;
    const bf16_t* cur;                                          // the slots that go on to LayerNorm + q

    // ---- this wave's weight stream: [W_ih][W_hh] [W1][W2] [Wq], blocks nb = w, w+4, ... (G3/16, H/16, D/16 are multiples of 4) ----
    static_assert(D == 192 && (G3 / 16) % 4 == 0 && (H / 16) % 4 == 0 && (D / 16) % 4 == 0 && H % 192 == 0, "block counts per wave");
    WStream ws;
    ws.total = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) ws.total += seg_at(a, i).nt * seg_at(a, i).nkc;
    ws.is = ws.it = ws.ikc = ws.issued = 0;
    while (ws.is < 5 && seg_at(a, ws.is).nt == 0) ++ws.is;
    const char* ring_ptr = sRing + w * (NSLOT * UNIT_BYTES);
    ws.ring = lds_addr_of(ring_ptr);
    int ucons = 0;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p) ws_issue(a, ws, w, lane);   // the first units land while the activations are loaded

    // ---- load the activation rows (16-byte pieces; rows past R are zero) ----
    {
        const bf16_t* U = static_cast<const bf16_t*>(a.upd);
        const bf16_t* Hs = static_cast<const bf16_t*>(a.h);
        for (int i = tid; i < ROWS * (D / 8); i += 256) {
            const int row = i / (D / 8), c8 = (i % (D / 8)) * 8;
            uint4 u = make_uint4(0, 0, 0, 0), h = u;
            if (r0 + row < R) {
                if (a.do_gru) u = *reinterpret_cast<const uint4*>(U + (int64_t)(r0 + row) * D + c8);
                h = *reinterpret_cast<const uint4*>(Hs + (int64_t)(r0 + row) * D + c8);
            }
            *reinterpret_cast<uint4*>(sU + row * PD + c8) = u;
            *reinterpret_cast<uint4*>(sHp + row * PD + c8) = h;
        }
    }
    __syncthreads();
    cur = sHp;
    if (a.do_gru) {
        // ---- gi = u W_ih^T + b_ih, gh = h W_hh^T + b_hh: stored (bias included, bf16) for the backward, gates from the stored values ----
        bf16_t* G = static_cast<bf16_t*>(a.g);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const bf16_t* W = static_cast<const bf16_t*>(which ? a.w_hh : a.w_ih);
            const float* bias = which ? a.b_hh : a.b_ih;
            bf16_t* sGw = sG + which * ROWS * G3;
            bf16_t* Gw = G + (int64_t)which * R * G3;
            (void)W;
            gemm16(a, ws, ucons, ring_ptr, which ? sHp : sU, PD, G3 / 64, 1, w, lane, [&](int col, int rbase, const f32x4& acc) {
                const float b = bias[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16_t v = f32_to_bf16(acc[r] + b);
                    sGw[(rbase + r) * G3 + col] = v;
                    if (r0 + rbase + r < R) Gw[(int64_t)(r0 + rbase + r) * G3 + col] = v;
                }
            });
        }
        __syncthreads();
        bf16_t* HN = static_cast<bf16_t*>(a.hn);
        for (int i = tid; i < ROWS * D; i += 256) {
            const int row = i / D, c = i % D;
            const bf16_t* ga = sG + row * G3;
            const bf16_t* gb = sG + ROWS * G3 + row * G3;
            const float rg = sigm_(bf16_to_f32(ga[c]) + bf16_to_f32(gb[c]));
            const float zg = sigm_(bf16_to_f32(ga[D + c]) + bf16_to_f32(gb[D + c]));
            const float ng = tanhf(bf16_to_f32(ga[2 * D + c]) + rg * bf16_to_f32(gb[2 * D + c]));
            const bf16_t v = f32_to_bf16((1.f - zg) * ng + zg * bf16_to_f32(sHp[row * PD + c]));
            sHn[row * PD + c] = v;
            if (r0 + row < R) HN[(int64_t)(r0 + row) * D + c] = v;
        }
        __syncthreads();
        cur = sHn;
        if (a.do_mlp) {
            ln16<D>(sHn, sY, PD, a.ln1_g, a.ln1_b, a.ln1_eps, static_cast<bf16_t*>(a.y), a.mean1, a.rstd1, r0, R, tid);
            __syncthreads();
            bf16_t* sA = sG;                                        // the gate tile is dead: [16][PH] hidden activations
            bf16_t* A = static_cast<bf16_t*>(a.a);
            gemm16(a, ws, ucons, ring_ptr, sY, PD, H / 64, 1, w, lane, [&](int col, int rbase, const f32x4& acc) {
                const float b = a.b1[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16_t v = f32_to_bf16(fmaxf(acc[r] + b, 0.f));
                    sA[(rbase + r) * PH + col] = v;
                    if (r0 + rbase + r < R) A[(int64_t)(r0 + rbase + r) * H + col] = v;
                }
            });
            __syncthreads();
            bf16_t* S = static_cast<bf16_t*>(a.s);
            gemm16(a, ws, ucons, ring_ptr, sA, PH, D / 64, H / 192, w, lane, [&](int col, int rbase, const f32x4& acc) {
                const float b = a.b2[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16_t o = f32_to_bf16(acc[r] + b + bf16_to_f32(sHn[(rbase + r) * PD + col]));
                    sU[(rbase + r) * PD + col] = o;
                    if (r0 + rbase + r < R) S[(int64_t)(r0 + rbase + r) * D + col] = o;
                }
            });
            __syncthreads();
            cur = sU;
        }
    }
    if (a.do_q) {
        ln16<D>(cur, sY, PD, a.ln2_g, a.ln2_b, a.ln2_eps, static_cast<bf16_t*>(a.sn), a.mean2, a.rstd2, r0, R, tid);
        __syncthreads();
        bf16_t* Q = static_cast<bf16_t*>(a.q);
        gemm16(a, ws, ucons, ring_ptr, sY, PD, D / 64, 1, w, lane, [&](int col, int rbase, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r0 + rbase + r < R) Q[(int64_t)(r0 + rbase + r) * D + col] = f32_to_bf16(acc[r]);
        });
    }
}

// ---- the same forward as a few right-sized launches ("staged") ---------------------------------------------------
// One workgroup per 16 rows streams 1 MB of weights through one CU (~29 us).  Cut at the points where a whole row is needed
// (the two LayerNorms and the K = 768 product), every piece splits over output-column tiles as well: 264 one-wave workgroups
// per launch, each loading a few KB of weights straight into MFMA fragments -- the kernel boundary is the cross-CU exchange.
//   gru  : (row block, 16 hidden units): the six gate tiles of those units, gate math, h'
//   mlp1 : (row block, 64 hidden columns): LayerNorm_mlp of its rows (redundantly per workgroup: 6 KB), fc1 + ReLU
//   mlp2 : (row block, 16 columns): fc2 over K = 768, + bias + h'
//   q    : (row block, 16 columns): LayerNorm_slots of its rows, q projection
// Same stored values and rounding points as the fused kernel; LayerNorm sums associate differently (4 lanes x 48 elements).
__device__ __forceinline__ bf16x8 frag16(const bf16_t* base, int64_t row, int64_t ld, int k) {
    return *reinterpret_cast<const bf16x8*>(base + row * ld + k);
}

// LayerNorm of the wave's 16 rows held as A fragments (lane: row = lane & 15, columns 32 ks + 8 kq .. + 7, ks < 6): the row's
// four lanes (kq = 0..3) combine with two shuffles.  Returns the normalised fragments; mean / rstd of the lane's row.
template <int D>
__device__ __forceinline__ void ln_frags(bf16x8 (&x)[D / 32], const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float eps, int kq, float& mean, float& rstd) {
    constexpr int KS = D / 32;
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[ks][e] = (float)x[ks][e]; s += v[ks][e]; }
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    mean = s * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dl = v[ks][e] - mean; q += dl * dl; }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    rstd = rsqrtf(q * (1.f / D) + eps);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ks * 32 + kq * 8 + e;
            x[ks][e] = (__bf16)((v[ks][e] - mean) * rstd * gamma[c] + beta[c]);
        }
}

template <int D>
__global__ __launch_bounds__(64) void tail_gru_kernel(const focus_slot_tail_args a) {
    constexpr int KS = D / 32, G3 = 3 * D;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int jt = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* U = static_cast<const bf16_t*>(a.upd);
    const bf16_t* Hs = static_cast<const bf16_t*>(a.h);
    const bf16_t* Wi = static_cast<const bf16_t*>(a.w_ih);
    const bf16_t* Wh = static_cast<const bf16_t*>(a.w_hh);
    bf16x8 au[KS], ah[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { au[ks] = frag16(U, row, D, ks * 32 + kq * 8); ah[ks] = frag16(Hs, row, D, ks * 32 + kq * 8); }
    f32x4 gi[3], gh[3];
    bf16x8 wi[3][KS], wh[3][KS];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int64_t n = g * D + jt * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { wi[g][ks] = frag16(Wi, n, D, ks * 32 + kq * 8); wh[g][ks] = frag16(Wh, n, D, ks * 32 + kq * 8); }
    }
    __builtin_amdgcn_sched_barrier(0);                  // all 48 fragment loads in flight together (see tail_mlp2_kernel)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(wi[g][ks])); asm volatile("" : "+v"(wh[g][ks])); }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        gi[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gh[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            gi[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(au[ks], wi[g][ks], gi[g], 0, 0, 0);
            gh[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wh[g][ks], gh[g], 0, 0, 0);
        }
    }
    // lane: rows kq*4 + r, hidden unit j
    const int j = jt * 16 + frow;
    bf16_t* G = static_cast<bf16_t*>(a.g);
    bf16_t* HN = static_cast<bf16_t*>(a.hn);
    float bi[3], bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bi[g] = a.b_ih[g * D + j]; bh[g] = a.b_hh[g * D + j]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + kq * 4 + r;
        if (rr >= R) continue;
        bf16_t vi[3], vh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            vi[g] = f32_to_bf16(gi[g][r] + bi[g]);
            vh[g] = f32_to_bf16(gh[g][r] + bh[g]);
            G[(int64_t)rr * G3 + g * D + j] = vi[g];
            G[((int64_t)R + rr) * G3 + g * D + j] = vh[g];
        }
        const float rg = sigm_(bf16_to_f32(vi[0]) + bf16_to_f32(vh[0]));
        const float zg = sigm_(bf16_to_f32(vi[1]) + bf16_to_f32(vh[1]));
        const float ng = tanhf(bf16_to_f32(vi[2]) + rg * bf16_to_f32(vh[2]));
        HN[(int64_t)rr * D + j] = f32_to_bf16((1.f - zg) * ng + zg * bf16_to_f32(Hs[(int64_t)rr * D + j]));
    }
}

template <int D, int H>
__global__ __launch_bounds__(64) void tail_mlp1_kernel(const focus_slot_tail_args a) {
    constexpr int KS = D / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* HN = static_cast<const bf16_t*>(a.hn);
    const bf16_t* W1 = static_cast<const bf16_t*>(a.w1);
    bf16x8 x[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) x[ks] = frag16(HN, row, D, ks * 32 + kq * 8);
    bf16x8 w[4][KS];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[t][ks] = frag16(W1, (grp * 4 + t) * 16 + frow, D, ks * 32 + kq * 8);
    float mean, rstd;
    ln_frags<D>(x, a.ln1_g, a.ln1_b, a.ln1_eps, kq, mean, rstd);
    if (grp == 0 && r0 + frow < R) {
        bf16_t* Y = static_cast<bf16_t*>(a.y);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(Y + (int64_t)(r0 + frow) * D + ks * 32 + kq * 8) = x[ks];
        if (kq == 0) { a.mean1[r0 + frow] = mean; a.rstd1[r0 + frow] = rstd; }
    }
    bf16_t* A = static_cast<bf16_t*>(a.a);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[ks], w[t][ks], acc, 0, 0, 0);
        const int col = (grp * 4 + t) * 16 + frow;
        const float b = a.b1[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + kq * 4 + r;
            if (rr < R) A[(int64_t)rr * H + col] = f32_to_bf16(fmaxf(acc[r] + b, 0.f));
        }
    }
}

template <int D, int H>
__global__ __launch_bounds__(64) void tail_mlp2_kernel(const focus_slot_tail_args a) {
    constexpr int KS = H / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int nt = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* A = static_cast<const bf16_t*>(a.a);
    const bf16_t* W2 = static_cast<const bf16_t*>(a.w2);
    const bf16_t* HN = static_cast<const bf16_t*>(a.hn);
    bf16x8 x[KS], w[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        x[ks] = frag16(A, row, H, ks * 32 + kq * 8);
        w[ks] = frag16(W2, nt * 16 + frow, H, ks * 32 + kq * 8);
    }
    const int col = nt * 16 + frow;
    float res[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) res[r] = bf16_to_f32(HN[(int64_t)min(r0 + kq * 4 + r, R - 1) * D + col]);
    // all 48 fragment loads in flight together (one round trip): left alone hipcc interleaves load and MFMA in 58 registers,
    // i.e. four dependent round trips for a 3 us kernel
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(x[ks])); asm volatile("" : "+v"(w[ks])); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[ks], w[ks], acc, 0, 0, 0);
    const float b = a.b2[col];
    bf16_t* S = static_cast<bf16_t*>(a.s);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + kq * 4 + r;
        if (rr < R) S[(int64_t)rr * D + col] = f32_to_bf16(acc[r] + b + res[r]);
    }
}

template <int D>
__global__ __launch_bounds__(64) void tail_q_kernel(const focus_slot_tail_args a, const bf16_t* __restrict__ cur) {
    constexpr int KS = D / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int nt = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* Wq = static_cast<const bf16_t*>(a.wq);
    bf16x8 x[KS], w[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { x[ks] = frag16(cur, row, D, ks * 32 + kq * 8); w[ks] = frag16(Wq, nt * 16 + frow, D, ks * 32 + kq * 8); }
    float mean, rstd;
    ln_frags<D>(x, a.ln2_g, a.ln2_b, a.ln2_eps, kq, mean, rstd);
    if (nt == 0 && r0 + frow < R) {
        bf16_t* SN = static_cast<bf16_t*>(a.sn);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(SN + (int64_t)(r0 + frow) * D + ks * 32 + kq * 8) = x[ks];
        if (kq == 0) { a.mean2[r0 + frow] = mean; a.rstd2[r0 + frow] = rstd; }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[ks], w[ks], acc, 0, 0, 0);
    bf16_t* Q = static_cast<bf16_t*>(a.q);
    const int col = nt * 16 + frow;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + kq * 4 + r;
        if (rr < R) Q[(int64_t)rr * D + col] = f32_to_bf16(acc[r]);
    }
}

template <int D, int H>
int launch_tail_staged(const focus_slot_tail_args& a, hipStream_t s) {
    const int rb = (a.R + ROWS - 1) / ROWS;
    const bf16_t* cur = static_cast<const bf16_t*>(a.h);
    if (a.do_gru) {
        hipLaunchKernelGGL((tail_gru_kernel<D>), dim3(D / 16, rb), dim3(64), 0, s, a);
        cur = static_cast<const bf16_t*>(a.hn);
        if (a.do_mlp) {
            hipLaunchKernelGGL((tail_mlp1_kernel<D, H>), dim3(H / 64, rb), dim3(64), 0, s, a);
            hipLaunchKernelGGL((tail_mlp2_kernel<D, H>), dim3(D / 16, rb), dim3(64), 0, s, a);
            cur = static_cast<const bf16_t*>(a.s);
        }
    }
    if (a.do_q) hipLaunchKernelGGL((tail_q_kernel<D>), dim3(D / 16, rb), dim3(64), 0, s, a, cur);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

// ---- backward -------------------------------------------------------------------------------------------------
// The same chain run backwards on the same 16 rows per workgroup, against the TRANSPOSED bf16 weight copies ([in][out]: the
// dX products are then the same act[16][KIN] . W'[NOUT][KIN]^T form and stream through the same ring):
//   dsn = dq Wq ; ds = LN_slots'(dsn) + dout ; dz = (ds W2) * (a > 0) ; dy1 = dz W1 ; dhn = LN_mlp'(dy1) + ds ;
//   GRU gates' -> dgi, dgh [3D] ; dupd = dgi W_ih ; dh = dgh W_hh + z * dhn
// Every dY a weight gradient needs (ds, dz, dgi, dgh; dq is an input) is written out once: the weight gradients themselves are
// formed later, stacked over all applications of the loop (ops.deferred_wgrads).  LayerNorm parameter gradients leave as one
// [2][workgroups][D] partial per LayerNorm.  Stored values round to bf16 exactly where the unfused kernels rounded.
__device__ __forceinline__ Seg seg_at(const focus_slot_tail_bwd_args& a, int i) {
    const int D = a.D, H = a.H;
    // (the index is laundered between the tests: as one chain on one value hipcc turns the five pointers into a table in
    // scratch memory indexed at run time)
    int j = __builtin_amdgcn_readfirstlane(i);
    Seg r = Seg{static_cast<const bf16_t*>(a.wq_t), D, a.do_q ? D / 64 : 0, 1};
    asm volatile("" : "+s"(j));
    if (j == 1) r = Seg{static_cast<const bf16_t*>(a.w2_t), D, a.do_mlp ? H / 64 : 0, 1};
    asm volatile("" : "+s"(j));
    if (j == 2) r = Seg{static_cast<const bf16_t*>(a.w1_t), H, a.do_mlp ? D / 64 : 0, H / 192};
    asm volatile("" : "+s"(j));
    if (j == 3) r = Seg{static_cast<const bf16_t*>(a.w_ih_t), 3 * D, a.do_gru ? D / 64 : 0, 3};
    asm volatile("" : "+s"(j));
    if (j == 4) r = Seg{static_cast<const bf16_t*>(a.w_hh_t), 3 * D, a.do_gru ? D / 64 : 0, 3};
    return r;
}

// LayerNorm backward of the 16 rows of LDS tiles (16 threads per row): dx = rstd * (dy g - mean(dy g) - xhat mean(dy g xhat))
// (+ res), written bf16 to sOut and gOut; then the column sums of dy xhat and dy over the rows into partial[0/1][block][D].
template <int D>
__device__ __forceinline__ void ln16_bwd(const bf16_t* sDy, const bf16_t* sX, const bf16_t* sRes, bf16_t* sOut, int pitch,
                                         const float* __restrict__ gamma, const float* sMean, const float* sRstd, bf16_t* gOut,
                                         float* partial, int nblk, int r0, int R, int tid) {
    constexpr int PER = D / 16;
    const int row = tid >> 4, l = tid & 15;
    const float mean = sMean[row], rstd = sRstd[row];
    float xh[PER], dg[PER];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = l + 16 * j;
        xh[j] = (bf16_to_f32(sX[row * pitch + c]) - mean) * rstd;
        dg[j] = bf16_to_f32(sDy[row * pitch + c]) * gamma[c];
        c1 += dg[j];
        c2 += dg[j] * xh[j];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    c1 *= 1.f / D; c2 *= 1.f / D;
    const bool live = r0 + row < R;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = l + 16 * j;
        float v = (dg[j] - c1 - xh[j] * c2) * rstd;
        if (sRes) v += bf16_to_f32(sRes[row * pitch + c]);
        const bf16_t o = f32_to_bf16(v);
        sOut[row * pitch + c] = o;
        if (live && gOut) gOut[(int64_t)(r0 + row) * D + c] = o;
    }
    // parameter-gradient partials: thread c < D sums its column over the 16 rows (rows past R hold dy = 0)
    if (tid < D) {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const float dy = bf16_to_f32(sDy[rr * pitch + tid]);
            sg += dy * (bf16_to_f32(sX[rr * pitch + tid]) - sMean[rr]) * sRstd[rr];
            sb += dy;
        }
        partial[(int64_t)blockIdx.x * D + tid] = sg;
        partial[(int64_t)(nblk + blockIdx.x) * D + tid] = sb;
    }
}

__device__ __forceinline__ void load_rows16(bf16_t* sDst, int pitch, const void* gsrc, int D, int r0, int R, int tid) {
    const bf16_t* G = static_cast<const bf16_t*>(gsrc);
    for (int i = tid; i < ROWS * (D / 8); i += 256) {
        const int row = i / (D / 8), c8 = (i % (D / 8)) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (G && r0 + row < R) v = *reinterpret_cast<const uint4*>(G + (int64_t)(r0 + row) * D + c8);
        *reinterpret_cast<uint4*>(sDst + row * pitch + c8) = v;
    }
}

template <int D, int H>
__global__ __launch_bounds__(256) void slot_tail_bwd_kernel(const focus_slot_tail_bwd_args a) {
    constexpr int PD = D + 8, PH = H + 8, G3 = 3 * D, PG = G3 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA0 = reinterpret_cast<bf16_t*>(smem);              // [16][PD]  dq, then dy1, then z * dhn
    bf16_t* sDs = sA0 + ROWS * PD;                              // [16][PD]  dout -> ds -> dhn
    bf16_t* sX = sDs + ROWS * PD;                               // [16][PD]  the LayerNorm inputs (slots, then h'), then h
    bf16_t* sT = sX + ROWS * PD;                                // [16][PD]  dsn
    bf16_t* sG = sT + ROWS * PD;                                // [2][16][PG] gates / gate gradients | [16][PH] dz
    float* sStat = reinterpret_cast<float*>(sG + 2 * ROWS * PG);   // [2][16] mean, rstd
    unsigned char* sMask = reinterpret_cast<unsigned char*>(sStat + 2 * ROWS);   // [16][H/8] bits: a > 0 (the ReLU's derivative)
    char* sRing = reinterpret_cast<char*>(sMask + ROWS * (H / 8));
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * ROWS, R = a.R, nblk = gridDim.x;
    static_assert(D == 192 && H % 192 == 0 && (H / 16) % 4 == 0, "block counts per wave");

    WStream ws;
    ws.total = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) ws.total += seg_at(a, i).nt * seg_at(a, i).nkc;
    ws.is = ws.it = ws.ikc = ws.issued = 0;
    while (ws.is < 5 && seg_at(a, ws.is).nt == 0) ++ws.is;
    const char* ring_ptr = sRing + w * (NSLOT * UNIT_BYTES);
    ws.ring = lds_addr_of(ring_ptr);
    int ucons = 0;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p) ws_issue(a, ws, w, lane);

    auto load_stats = [&](const float* m, const float* r) __attribute__((always_inline)) {
        if (tid < ROWS) {
            const bool live = r0 + tid < R;
            sStat[tid] = live ? m[r0 + tid] : 0.f;
            sStat[ROWS + tid] = live ? r[r0 + tid] : 0.f;
        }
    };
    load_rows16(sDs, PD, a.dout, D, r0, R, tid);                 // (NULL: zeros)
    if (a.do_mlp) {
        // the ReLU mask as bits, read now: a global load inside the products' epilogues would drain the weight ring
        const bf16_t* A = static_cast<const bf16_t*>(a.a);
        for (int i = tid; i < ROWS * (H / 8); i += 256) {
            const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
            unsigned m = 0;
            if (r0 + row < R) {
                const uint4 v = *reinterpret_cast<const uint4*>(A + (int64_t)(r0 + row) * H + c8);
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t bits = (wv[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                    m |= (bits != 0 && !(bits & 0x8000u)) ? (1u << e) : 0u;   // a = relu(..) >= 0: positive <=> non-zero, sign clear
                }
            }
            sMask[i] = (unsigned char)m;
        }
    }
    if (a.do_q) {
        load_rows16(sA0, PD, a.dq, D, r0, R, tid);
        load_rows16(sX, PD, a.cur, D, r0, R, tid);
        load_stats(a.mean2, a.rstd2);
        __syncthreads();
        // dsn = dq . Wq
        gemm16(a, ws, ucons, ring_ptr, sA0, PD, D / 64, 1, w, lane, [&](int col, int rbase, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sT[(rbase + r) * PD + col] = f32_to_bf16(acc[r]);
        });
        __syncthreads();
        ln16_bwd<D>(sT, sX, sDs, sDs, PD, a.ln2_g, sStat, sStat + ROWS, static_cast<bf16_t*>(a.ds), a.part2, nblk, r0, R, tid);
    } else if (a.ds) {
        // no query this time: ds = dout (still written: it is the dY of fc2's weight gradient)
        __syncthreads();
        bf16_t* DS = static_cast<bf16_t*>(a.ds);
        for (int i = tid; i < ROWS * D; i += 256) {
            const int row = i / D, c = i % D;
            if (r0 + row < R) DS[(int64_t)(r0 + row) * D + c] = sDs[row * PD + c];
        }
    }
    __syncthreads();
    if (a.do_mlp) {
        // dz = (ds . W2) * (a > 0)
        bf16_t* sDz = sG;
        bf16_t* DZ = static_cast<bf16_t*>(a.dz);
        gemm16(a, ws, ucons, ring_ptr, sDs, PD, H / 64, 1, w, lane, [&](int col, int rbase, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = r0 + rbase + r < R;
                const bool on = (sMask[(rbase + r) * (H / 8) + (col >> 3)] >> (col & 7)) & 1;
                const bf16_t v = f32_to_bf16(on ? acc[r] : 0.f);
                sDz[(rbase + r) * PH + col] = v;
                if (live) DZ[(int64_t)(r0 + rbase + r) * H + col] = v;
            }
        });
        load_rows16(sX, PD, a.hn, D, r0, R, tid);               // (sX is free: LN_slots' is done)
        load_stats(a.mean1, a.rstd1);
        __syncthreads();
        // dy1 = dz . W1
        gemm16(a, ws, ucons, ring_ptr, sDz, PH, D / 64, H / 192, w, lane, [&](int col, int rbase, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sA0[(rbase + r) * PD + col] = f32_to_bf16(acc[r]);
        });
        __syncthreads();
        ln16_bwd<D>(sA0, sX, sDs, sDs, PD, a.ln1_g, sStat, sStat + ROWS, nullptr, a.part1, nblk, r0, R, tid);
        __syncthreads();
    }
    if (a.do_gru) {
        // gate gradients from the stored pre-activations (bias included), h and dhn; in place in the padded gate tile
        const bf16_t* G = static_cast<const bf16_t*>(a.g);
        for (int i = tid; i < 2 * ROWS * (G3 / 8); i += 256) {
            const int which = i / (ROWS * (G3 / 8)), rem = i % (ROWS * (G3 / 8));
            const int row = rem / (G3 / 8), c8 = (rem % (G3 / 8)) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r0 + row < R) v = *reinterpret_cast<const uint4*>(G + ((int64_t)which * R + r0 + row) * G3 + c8);
            *reinterpret_cast<uint4*>(sG + (which * ROWS + row) * PG + c8) = v;
        }
        load_rows16(sX, PD, a.h, D, r0, R, tid);
        __syncthreads();
        bf16_t* DG = static_cast<bf16_t*>(a.dg);
        for (int i = tid; i < ROWS * D; i += 256) {
            const int row = i / D, c = i % D;
            bf16_t* ga = sG + row * PG;
            bf16_t* gb = sG + (ROWS + row) * PG;
            const float hn_ = bf16_to_f32(gb[2 * D + c]);
            const float rg = sigm_(bf16_to_f32(ga[c]) + bf16_to_f32(gb[c]));
            const float zg = sigm_(bf16_to_f32(ga[D + c]) + bf16_to_f32(gb[D + c]));
            const float ng = tanhf(bf16_to_f32(ga[2 * D + c]) + rg * hn_);
            const float hv = bf16_to_f32(sX[row * PD + c]), gd = bf16_to_f32(sDs[row * PD + c]);
            const float dn = gd * (1.f - zg), dzg = gd * (hv - ng);
            const float dpre_n = dn * (1.f - ng * ng);
            const float dr = dpre_n * hn_;
            const bf16_t pr = f32_to_bf16(dr * rg * (1.f - rg)), pz = f32_to_bf16(dzg * zg * (1.f - zg));
            const bf16_t pn = f32_to_bf16(dpre_n), pnr = f32_to_bf16(dpre_n * rg);
            ga[c] = pr; ga[D + c] = pz; ga[2 * D + c] = pn;
            gb[c] = pr; gb[D + c] = pz; gb[2 * D + c] = pnr;
            sA0[row * PD + c] = f32_to_bf16(gd * zg);            // the direct path of dh
            if (r0 + row < R) {
                bf16_t* o0 = DG + (int64_t)(r0 + row) * G3;
                bf16_t* o1 = DG + ((int64_t)R + r0 + row) * G3;
                o0[c] = pr; o0[D + c] = pz; o0[2 * D + c] = pn;
                o1[c] = pr; o1[D + c] = pz; o1[2 * D + c] = pnr;
            }
        }
        __syncthreads();
        bf16_t* DU = static_cast<bf16_t*>(a.dupd);
        bf16_t* DH = static_cast<bf16_t*>(a.dh);
        gemm16(a, ws, ucons, ring_ptr, sG, PG, D / 64, 3, w, lane, [&](int col, int rbase, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r0 + rbase + r < R) DU[(int64_t)(r0 + rbase + r) * D + col] = f32_to_bf16(acc[r]);
        });
        gemm16(a, ws, ucons, ring_ptr, sG + ROWS * PG, PG, D / 64, 3, w, lane, [&](int col, int rbase, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r0 + rbase + r < R)
                    DH[(int64_t)(r0 + rbase + r) * D + col] = f32_to_bf16(acc[r] + bf16_to_f32(sA0[(rbase + r) * PD + col]));
        });
    } else {
        bf16_t* DH = static_cast<bf16_t*>(a.dh);
        for (int i = tid; i < ROWS * D; i += 256) {
            const int row = i / D, c = i % D;
            if (r0 + row < R) DH[(int64_t)(r0 + row) * D + c] = sDs[row * PD + c];
        }
    }
}

// ---- the backward as five right-sized launches ---------------------------------------------------------------
//   bq   : (row block, 16 columns)        dsn = dq . Wq
//   bln2 : (row block, 64 hidden columns) ds = LN_slots'(dsn) + dout (redundantly per workgroup), dz = (ds . W2) * (a > 0)
//   bfc1 : (row block, 16 columns)        dy1 = dz . W1   (K = 768)
//   bgate: (row block, 16 hidden units)   dhn = LN_mlp'(dy1) + ds (redundantly), gate gradients of those units, z * dhn
//   bgru : (2 x row block x 16 columns)   dupd = dgi . W_ih ;  dh = dgh . W_hh + z * dhn
// Same outputs as slot_tail_bwd_kernel (+ three [R, D] scratch matrices between the launches).
template <int D>
__device__ __forceinline__ void ln_bwd_frags(const bf16x8 (&dy)[D / 32], const bf16x8 (&x)[D / 32], const bf16x8* res,
                                             const float* __restrict__ gamma, float mean, float rstd, int kq, bool live, int frow,
                                             bf16x8 (&out)[D / 32], float* partial_g, float* partial_b) {
    constexpr int KS = D / 32;
    float xh[KS][8], dg[KS][8];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh[ks][e] = ((float)x[ks][e] - mean) * rstd;
            dg[ks][e] = (float)dy[ks][e] * gamma[ks * 32 + kq * 8 + e];
            c1 += dg[ks][e];
            c2 += dg[ks][e] * xh[ks][e];
        }
    c1 += __shfl_xor(c1, 16, 64); c1 += __shfl_xor(c1, 32, 64);
    c2 += __shfl_xor(c2, 16, 64); c2 += __shfl_xor(c2, 32, 64);
    c1 *= 1.f / D; c2 *= 1.f / D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (dg[ks][e] - c1 - xh[ks][e] * c2) * rstd;
            if (res) v += (float)res[ks][e];
            out[ks][e] = (__bf16)v;
        }
    if (partial_g) {
        // column sums over the wave's 16 rows (the lanes that share kq): rows past R contribute nothing
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = live ? (float)dy[ks][e] * xh[ks][e] : 0.f, b = live ? (float)dy[ks][e] : 0.f;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (frow == 0) { partial_g[ks * 32 + kq * 8 + e] = a; partial_b[ks * 32 + kq * 8 + e] = b; }
            }
    }
}

template <int D>
__global__ __launch_bounds__(64) void tailb_q_kernel(const focus_slot_tail_bwd_args a) {
    constexpr int KS = D / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int nt = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* DQ = static_cast<const bf16_t*>(a.dq);
    const bf16_t* W = static_cast<const bf16_t*>(a.wq_t);
    bf16x8 x[KS], w[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { x[ks] = frag16(DQ, row, D, ks * 32 + kq * 8); w[ks] = frag16(W, nt * 16 + frow, D, ks * 32 + kq * 8); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[ks], w[ks], acc, 0, 0, 0);
    bf16_t* O = static_cast<bf16_t*>(a.ws_dsn);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + kq * 4 + r;
        if (rr < R) O[(int64_t)rr * D + nt * 16 + frow] = f32_to_bf16(acc[r]);
    }
}

template <int D, int H>
__global__ __launch_bounds__(64) void tailb_ln2_kernel(const focus_slot_tail_bwd_args a) {
    constexpr int KS = D / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R, nblk = gridDim.y;
    const bool live = r0 + frow < R;
    const int64_t row = min(r0 + frow, R - 1);
    bf16x8 ds[KS];
    {
        const bf16_t* DO = static_cast<const bf16_t*>(a.dout);
        bf16x8 res[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (DO) res[ks] = frag16(DO, row, D, ks * 32 + kq * 8);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) res[ks][e] = (__bf16)0.f;
        }
        if (a.do_q) {
            const bf16_t* DSN = static_cast<const bf16_t*>(a.ws_dsn);
            const bf16_t* CUR = static_cast<const bf16_t*>(a.cur);
            bf16x8 dy[KS], x[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { dy[ks] = frag16(DSN, row, D, ks * 32 + kq * 8); x[ks] = frag16(CUR, row, D, ks * 32 + kq * 8); }
            const bool first = grp == 0;
            ln_bwd_frags<D>(dy, x, res, a.ln2_g, a.mean2[row], a.rstd2[row], kq, live, frow, ds,
                            first ? a.part2 + (int64_t)blockIdx.y * D : nullptr,
                            first ? a.part2 + (int64_t)(nblk + blockIdx.y) * D : nullptr);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ds[ks] = res[ks];
        }
    }
    if (grp == 0 && live && a.ds) {
        bf16_t* DS = static_cast<bf16_t*>(a.ds);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(DS + (int64_t)(r0 + frow) * D + ks * 32 + kq * 8) = ds[ks];
    }
    if (!a.do_mlp) return;
    const bf16_t* W = static_cast<const bf16_t*>(a.w2_t);
    const bf16_t* A = static_cast<const bf16_t*>(a.a);
    bf16_t* DZ = static_cast<bf16_t*>(a.dz);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = (grp * 4 + t) * 16 + frow;
        bf16x8 w[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[ks] = frag16(W, col, D, ks * 32 + kq * 8);
        bf16_t av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = A[(int64_t)min(r0 + kq * 4 + r, R - 1) * H + col];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ds[ks], w[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + kq * 4 + r;
            if (rr < R) DZ[(int64_t)rr * H + col] = f32_to_bf16(bf16_to_f32(av[r]) > 0.f ? acc[r] : 0.f);
        }
    }
}

template <int D, int H>
__global__ __launch_bounds__(64) void tailb_fc1_kernel(const focus_slot_tail_bwd_args a) {
    constexpr int KS = H / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int nt = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* DZ = static_cast<const bf16_t*>(a.dz);
    const bf16_t* W = static_cast<const bf16_t*>(a.w1_t);
    bf16x8 x[KS], w[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { x[ks] = frag16(DZ, row, H, ks * 32 + kq * 8); w[ks] = frag16(W, nt * 16 + frow, H, ks * 32 + kq * 8); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(x[ks])); asm volatile("" : "+v"(w[ks])); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[ks], w[ks], acc, 0, 0, 0);
    bf16_t* O = static_cast<bf16_t*>(a.ws_dy1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + kq * 4 + r;
        if (rr < R) O[(int64_t)rr * D + nt * 16 + frow] = f32_to_bf16(acc[r]);
    }
}

template <int D>
__global__ __launch_bounds__(64) void tailb_gate_kernel(const focus_slot_tail_bwd_args a) {
    constexpr int KS = D / 32, G3 = 3 * D, PD = D + 8;
    __shared__ __attribute__((aligned(16))) bf16_t sD[ROWS * PD];          // dhn rows: A-fragment layout in, (row, unit) layout out
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int jt = blockIdx.x, r0 = blockIdx.y * ROWS, R = a.R, nblk = gridDim.y;
    const bool live = r0 + frow < R;
    const int64_t row = min(r0 + frow, R - 1);
    bf16x8 dhn[KS];
    {
        const bf16_t* DS = static_cast<const bf16_t*>(a.ds ? a.ds : a.dout);       // the gradient arriving at h'
        bf16x8 res[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (DS) res[ks] = frag16(DS, row, D, ks * 32 + kq * 8);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) res[ks][e] = (__bf16)0.f;
        }
        if (a.do_mlp) {
            const bf16_t* DY = static_cast<const bf16_t*>(a.ws_dy1);
            const bf16_t* HN = static_cast<const bf16_t*>(a.hn);
            bf16x8 dy[KS], x[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { dy[ks] = frag16(DY, row, D, ks * 32 + kq * 8); x[ks] = frag16(HN, row, D, ks * 32 + kq * 8); }
            const bool first = jt == 0;
            ln_bwd_frags<D>(dy, x, res, a.ln1_g, a.mean1[row], a.rstd1[row], kq, live, frow, dhn,
                            first ? a.part1 + (int64_t)blockIdx.y * D : nullptr,
                            first ? a.part1 + (int64_t)(nblk + blockIdx.y) * D : nullptr);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dhn[ks] = res[ks];
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(sD + frow * PD + ks * 32 + kq * 8) = dhn[ks];
    __syncthreads();
    const int j = jt * 16 + frow;
    const bf16_t* G = static_cast<const bf16_t*>(a.g);
    const bf16_t* Hs = static_cast<const bf16_t*>(a.h);
    bf16_t* DG = static_cast<bf16_t*>(a.dg);
    bf16_t* RES = static_cast<bf16_t*>(a.ws_res);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rl = kq * 4 + r, rr = r0 + rl;
        if (rr >= R) continue;
        const bf16_t* ga = G + (int64_t)rr * G3;
        const bf16_t* gb = G + ((int64_t)R + rr) * G3;
        const float hn_ = bf16_to_f32(gb[2 * D + j]);
        const float rg = sigm_(bf16_to_f32(ga[j]) + bf16_to_f32(gb[j]));
        const float zg = sigm_(bf16_to_f32(ga[D + j]) + bf16_to_f32(gb[D + j]));
        const float ng = tanhf(bf16_to_f32(ga[2 * D + j]) + rg * hn_);
        const float hv = bf16_to_f32(Hs[(int64_t)rr * D + j]), gd = bf16_to_f32(sD[rl * PD + j]);
        const float dn = gd * (1.f - zg), dzg = gd * (hv - ng);
        const float dpre_n = dn * (1.f - ng * ng);
        const float dr = dpre_n * hn_;
        const bf16_t pr = f32_to_bf16(dr * rg * (1.f - rg)), pz = f32_to_bf16(dzg * zg * (1.f - zg));
        const bf16_t pn = f32_to_bf16(dpre_n), pnr = f32_to_bf16(dpre_n * rg);
        bf16_t* o0 = DG + (int64_t)rr * G3;
        bf16_t* o1 = DG + ((int64_t)R + rr) * G3;
        o0[j] = pr; o0[D + j] = pz; o0[2 * D + j] = pn;
        o1[j] = pr; o1[D + j] = pz; o1[2 * D + j] = pnr;
        RES[(int64_t)rr * D + j] = f32_to_bf16(gd * zg);
    }
}

template <int D>
__global__ __launch_bounds__(64) void tailb_gru_kernel(const focus_slot_tail_bwd_args a) {
    constexpr int G3 = 3 * D, KS = G3 / 32;
    const int lane = threadIdx.x, frow = lane & 15, kq = lane >> 4;
    const int which = blockIdx.x / (D / 16), nt = blockIdx.x % (D / 16), r0 = blockIdx.y * ROWS, R = a.R;
    const int64_t row = min(r0 + frow, R - 1);
    const bf16_t* DG = static_cast<const bf16_t*>(a.dg) + (int64_t)which * R * G3;
    const bf16_t* W = static_cast<const bf16_t*>(which ? a.w_hh_t : a.w_ih_t);
    bf16x8 x[KS], w[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { x[ks] = frag16(DG, row, G3, ks * 32 + kq * 8); w[ks] = frag16(W, nt * 16 + frow, G3, ks * 32 + kq * 8); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(x[ks])); asm volatile("" : "+v"(w[ks])); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[ks], w[ks], acc, 0, 0, 0);
    bf16_t* O = static_cast<bf16_t*>(which ? a.dh : a.dupd);
    const bf16_t* RES = static_cast<const bf16_t*>(a.ws_res);
    const int col = nt * 16 + frow;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + kq * 4 + r;
        if (rr < R) O[(int64_t)rr * D + col] = f32_to_bf16(acc[r] + (which ? bf16_to_f32(RES[(int64_t)rr * D + col]) : 0.f));
    }
}

template <int D, int H>
int launch_tail_bwd_staged(const focus_slot_tail_bwd_args& a, hipStream_t s) {
    const int rb = (a.R + ROWS - 1) / ROWS;
    if (a.do_q) hipLaunchKernelGGL((tailb_q_kernel<D>), dim3(D / 16, rb), dim3(64), 0, s, a);
    if (a.do_q || a.do_mlp) hipLaunchKernelGGL((tailb_ln2_kernel<D, H>), dim3(a.do_mlp ? H / 64 : 1, rb), dim3(64), 0, s, a);
    if (a.do_mlp) hipLaunchKernelGGL((tailb_fc1_kernel<D, H>), dim3(D / 16, rb), dim3(64), 0, s, a);
    if (a.do_gru) {
        hipLaunchKernelGGL((tailb_gate_kernel<D>), dim3(D / 16, rb), dim3(64), 0, s, a);
        hipLaunchKernelGGL((tailb_gru_kernel<D>), dim3(2 * (D / 16), rb), dim3(64), 0, s, a);
    }
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <int D, int H>
int launch_tail_bwd(const focus_slot_tail_bwd_args& a, hipStream_t s) {
    constexpr size_t lds = (size_t)(4 * ROWS * (D + 8) + 2 * ROWS * (3 * D + 8)) * 2 + 2 * ROWS * 4 + ROWS * (H / 8) +
                           (size_t)4 * NSLOT * UNIT_BYTES;
    static_assert(2 * ROWS * (3 * D + 8) >= ROWS * (H + 8), "the dz tile reuses the gate tile");
    static_assert(lds <= 160 * 1024, "LDS");
    auto k = slot_tail_bwd_kernel<D, H>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(k, dim3((a.R + ROWS - 1) / ROWS), dim3(256), lds, s, a);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <int D, int H>
int launch_tail_fwd(const focus_slot_tail_args& a, hipStream_t s) {
    constexpr size_t lds = (size_t)(4 * ROWS * (D + 8) + 2 * ROWS * 3 * D) * 2 + (size_t)4 * NSLOT * UNIT_BYTES;
    static_assert(2 * ROWS * 3 * D >= ROWS * (H + 8), "the MLP hidden tile reuses the gate tile");
    auto k = slot_tail_fwd_kernel<D, H>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(k, dim3((a.R + ROWS - 1) / ROWS), dim3(256), lds, s, a);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

}  // namespace

extern "C" int focus_slot_tail_ok(int D, int H, int dtype) { return dtype == FOCUS_BF16 && D == 192 && H == 768; }

extern "C" int focus_slot_tail_fwd(const focus_slot_tail_args* args, void* stream) {
    if (!args) return FOCUS_ERR_NULL;
    const focus_slot_tail_args& a = *args;
    if (a.R <= 0) return FOCUS_OK;
    if (!focus_slot_tail_ok(a.D, a.H, FOCUS_BF16)) return FOCUS_ERR_SHAPE;
    if (!a.h || (a.do_gru && (!a.upd || !a.w_ih || !a.w_hh || !a.b_ih || !a.b_hh || !a.g || !a.hn))) return FOCUS_ERR_NULL;
    if (a.do_mlp && (!a.do_gru || !a.ln1_g || !a.ln1_b || !a.w1 || !a.b1 || !a.w2 || !a.b2 || !a.y || !a.mean1 || !a.rstd1 || !a.a || !a.s))
        return FOCUS_ERR_NULL;
    if (a.do_q && (!a.ln2_g || !a.ln2_b || !a.wq || !a.sn || !a.mean2 || !a.rstd2 || !a.q)) return FOCUS_ERR_NULL;
    // the right-sized launches by default (STEVE slot update, whole-step graph: 17.5 -> 16.2 ms; graphed product loop 20.5 ->
    // 19.2 ms; eager unchanged within its noise); FOCUS_SLOT_TAIL_STAGED=0: the one-workgroup-per-16-rows kernel
    const char* env = getenv("FOCUS_SLOT_TAIL_STAGED");             // (read per call: tests compare the two forms in one process)
    const int staged = env ? atoi(env) : 1;
    if (staged) return launch_tail_staged<192, 768>(a, static_cast<hipStream_t>(stream));
    return launch_tail_fwd<192, 768>(a, static_cast<hipStream_t>(stream));
}

extern "C" int focus_slot_tail_bwd_blocks(int R) { return (R + ROWS - 1) / ROWS; }

extern "C" int focus_slot_tail_bwd(const focus_slot_tail_bwd_args* args, void* stream) {
    if (!args) return FOCUS_ERR_NULL;
    const focus_slot_tail_bwd_args& a = *args;
    if (a.R <= 0) return FOCUS_OK;
    if (!focus_slot_tail_ok(a.D, a.H, FOCUS_BF16)) return FOCUS_ERR_SHAPE;
    if (!a.dh) return FOCUS_ERR_NULL;
    if (a.do_q && (!a.dq || !a.cur || !a.mean2 || !a.rstd2 || !a.ln2_g || !a.wq_t || !a.ds || !a.part2)) return FOCUS_ERR_NULL;
    if (a.do_mlp && (!a.do_gru || !a.a || !a.hn || !a.mean1 || !a.rstd1 || !a.ln1_g || !a.w1_t || !a.w2_t || !a.ds || !a.dz || !a.part1))
        return FOCUS_ERR_NULL;
    if (a.do_gru && (!a.g || !a.h || !a.w_ih_t || !a.w_hh_t || !a.dg || !a.dupd)) return FOCUS_ERR_NULL;
    const char* env = getenv("FOCUS_SLOT_TAIL_STAGED");             // 1 (default): five right-sized launches; 0: one launch
    if (env ? atoi(env) : 1) {
        if ((a.do_q && !a.ws_dsn) || (a.do_mlp && !a.ws_dy1) || (a.do_gru && !a.ws_res)) return FOCUS_ERR_WORKSPACE;
        if (!a.do_gru) {
            // the query-only application (a frame's first call): dh = LN_slots'(dq . Wq) + dout, written where ds would go
            if (!a.do_q) return FOCUS_ERR_SHAPE;
            focus_slot_tail_bwd_args b = a;
            b.ds = a.dh;
            return launch_tail_bwd_staged<192, 768>(b, static_cast<hipStream_t>(stream));
        }
        return launch_tail_bwd_staged<192, 768>(a, static_cast<hipStream_t>(stream));
    }
    return launch_tail_bwd<192, 768>(a, static_cast<hipStream_t>(stream));
}
