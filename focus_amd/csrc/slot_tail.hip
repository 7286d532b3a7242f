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

// DMA of the next unit of the stream into slot (issued % NSLOT); lane -> 16-byte chunk c = 64 p + lane of the [16][24] chunk
// grid; the chunk a row keeps at position j of an 8-chunk group is chunk j ^ ((row >> 1) & 7) (conflict-free fragment reads)
__device__ __forceinline__ void ws_issue(const focus_slot_tail_args& a, WStream& ws, int w, int lane) {
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
template <typename Epi>
__device__ __forceinline__ void gemm16(const focus_slot_tail_args& a, WStream& ws, int& ucons, const char* ring_ptr, const bf16_t* sAct,
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
    return launch_tail_fwd<192, 768>(a, static_cast<hipStream_t>(stream));
}
