// misc.hip -- small HBM-bound kernels around the GEMMs: bias-gradient column sums, dtype casts,
// transposed (zero-padded) operand copies, diagonal gather/scatter, patch im2col, token assembly,
// label-smoothing cross-entropy, GRU gate math, per-RoI cell max.
#include "focus_common.h"
#include "focus_debug.h"

namespace {

// ---- column sum: out[n] = sum_m x[m,n] ------------------------------------------------------------
// grid (ceil(N/64), MB): each block sums a strip of rows for 64 columns, 256 threads = 4 row-lanes x 64.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     int64_t ld_, int rows_per_block, int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r4 = threadIdx.x >> 6;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int m = m_begin + r4; m < m_end; m += 4) s += ld<T>(x + (int64_t)m * ld_ + c);
    red[r4][threadIdx.x & 63] = s;
    __syncthreads();
    if (r4 == 0 && c < N) {
        const int l = threadIdx.x;
        const float t = red[0][l] + red[1][l] + red[2][l] + red[3][l];
        if (gridDim.y == 1 && !accumulate) out[c] = t; else atomicAdd(out + c, t);
    }
}

// 16-byte loads: thread (tx, ty) sums columns 8*tx..8*tx+7 over rows ty, ty+8, ... of its strip
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* __restrict__ x, float* __restrict__ out,
                                                              int M, int N, int64_t ld_, int rows_per_block, int accumulate) {
    __shared__ float red[8][32][9];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + tx) * 8;
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N)
        for (int m = m_begin + ty; m < m_end; m += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + (int64_t)m * ld_ + c);
            s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
            s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
            s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
            s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ty][tx][j] = s[j];
    __syncthreads();
    const int j = threadIdx.x & 7, t2 = threadIdx.x >> 3;     // 32 column groups x 8 columns
    const int cc = (blockIdx.x * 32 + t2) * 8 + j;
    if (cc < N) {
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 8; ++y) t += red[y][t2][j];
        if (gridDim.y == 1 && !accumulate) out[cc] = t; else atomicAdd(out + cc, t);
    }
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t step = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += step) st4<TD>(d + i, ld4<TS>(s + i));
    if (i < n && i + 3 >= n)
        for (int64_t j = i; j < n; ++j) st<TD>(d + j, ld<TS>(s + j));
}

// ---- transpose with zero padding: dst[c][r] = src[r][c], dst row length dst_ld (>= R) ---------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const TS* __restrict__ src, int64_t src_ld,
                                                            int64_t src_bs, TD* __restrict__ dst, int64_t dst_ld,
                                                            int64_t dst_bs, int R, int Cc) {
    __shared__ float tile[64][65];
    const TS* s = src + (int64_t)blockIdx.z * src_bs;
    TD* d = dst + (int64_t)blockIdx.z * dst_bs;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? ld<TS>(s + (int64_t)r * src_ld + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < dst_ld) st<TD>(d + (int64_t)c * dst_ld + r, tile[tx][i]);
    }
}

template <typename T>
__global__ void diag_gather_kernel(const T* __restrict__ xt, T* __restrict__ xd, int64_t rows, int S, int F, int P,
                                   int C) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * C) return;
    const int64_t row = i / C;  // b*S + s
    const int c = (int)(i % C), s = (int)(row % S), f = s / P;
    st4<T>(xd + i, ld4<T>(xt + (row * F + f) * C + c));
}
template <typename T>
__global__ void diag_scatter_add_kernel(const T* __restrict__ dxd, T* __restrict__ dxt, int64_t rows, int S, int F,
                                        int P, int C) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * C) return;
    const int64_t row = i / C;
    const int c = (int)(i % C), s = (int)(row % S), f = s / P;
    T* p = dxt + (row * F + f) * C + c;
    const f4 a = ld4<T>(p), b = ld4<T>(dxd + i);
    st4<T>(p, (f4){a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w});
}

// ---- im2col for kernel==stride patches --------------------------------------------------------------
// one thread per 4 consecutive dw of one (token, c, dt, dh): reads 16 B of fp32, writes 4 elements.
template <typename T>
__global__ void im2col_kernel(const float* __restrict__ x, T* __restrict__ cols, int B, int Cin, int Tn, int H, int W,
                              int kt, int kh, int kw) {
    const int To = Tn / kt, Ho = H / kh, Wo = W / kw;
    const int Kc = Cin * kt * kh * kw;
    const int64_t total = (int64_t)B * To * Ho * Wo * Kc / 4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kq = (int)(i % (Kc / 4)) * 4;
    const int64_t tok = i / (Kc / 4);
    const int dw = kq % kw, dh = (kq / kw) % kh, dt = (kq / (kw * kh)) % kt, c = kq / (kw * kh * kt);
    const int wo = (int)(tok % Wo), ho = (int)((tok / Wo) % Ho), to = (int)((tok / ((int64_t)Wo * Ho)) % To);
    const int b = (int)(tok / ((int64_t)Wo * Ho * To));
    const float* p = x + ((((int64_t)b * Cin + c) * Tn + (to * kt + dt)) * H + (ho * kh + dh)) * W + wo * kw + dw;
    st4<T>(cols + tok * Kc + kq, ld4<float>(p));
}

template <typename T>
__global__ void embed_assemble_kernel(const T* __restrict__ patch, const float* __restrict__ cls,
                                      const float* __restrict__ pos, const float* __restrict__ temp,
                                      T* __restrict__ tok, int B, int Tn, int P, int C) {
    const int64_t N = 1 + (int64_t)Tn * P;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (int64_t)B * N * C) return;
    const int c = (int)(i % C);
    const int64_t row = i / C;
    const int n = (int)(row % N), b = (int)(row / N);
    f4 o;
    if (n == 0) {
        const f4 a = ld4<float>(cls + c), p0 = ld4<float>(pos + c);
        o = {a.x + p0.x, a.y + p0.y, a.z + p0.z, a.w + p0.w};
    } else {
        const int t = (n - 1) / P, p = (n - 1) % P;
        const f4 a = ld4<T>(patch + ((int64_t)b * Tn * P + (n - 1)) * C + c);
        const f4 ps = ld4<float>(pos + (int64_t)(1 + p) * C + c), te = ld4<float>(temp + (int64_t)t * C + c);
        o = {a.x + ps.x + te.x, a.y + ps.y + te.y, a.z + ps.z + te.z, a.w + ps.w + te.w};
    }
    st4<T>(tok + i, o);
}

// ---- label smoothing cross entropy: one block (256 threads) per row -------------------------------
__global__ __launch_bounds__(256) void xent_ls_kernel(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ target,
                                                      float* __restrict__ loss_rows, float* __restrict__ dlogits, int R,
                                                      int Ncls, float smoothing) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const float* x = logits + (int64_t)r * Ncls;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < Ncls; i += 256) m = fmaxf(m, x[i]);
    m = block_max(m, red);
    float s = 0.f, sx = 0.f;
    for (int i = threadIdx.x; i < Ncls; i += 256) { s += expf(x[i] - m); sx += x[i]; }
    s = block_sum(s, red);
    sx = block_sum(sx, red);
    const float lse = m + logf(s);
    const int t = (int)target[r];
    const float conf = 1.f - smoothing;
    if (threadIdx.x == 0) {
        const float nll = lse - x[t];
        const float smooth = lse - sx / (float)Ncls;
        loss_rows[r] = conf * nll + smoothing * smooth;
    }
    // d(mean over rows)/dlogits = (softmax - conf*onehot - smoothing/Ncls) / R
    for (int i = threadIdx.x; i < Ncls; i += 256) {
        const float p = expf(x[i] - lse);
        dlogits[(int64_t)r * Ncls + i] = (p - (i == t ? conf : 0.f) - smoothing / (float)Ncls) / (float)R;
    }
}

// ---- GRU gates --------------------------------------------------------------------------------------
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
// b_ih / b_hh (fp32 [3D], both or neither): the gate pre-activations arrive WITHOUT their biases (the two Linear products
// of the cell run as one batched bias-free launch); they are added here and the biased values written back, so that
// gi / gh hold what the backward kernel expects.
template <typename T>
__global__ void gru_fwd_kernel(T* __restrict__ gi, T* __restrict__ gh, const T* __restrict__ h, T* __restrict__ hn,
                               const float* __restrict__ b_ih, const float* __restrict__ b_hh, int R, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * D) return;
    const int64_t r = i / D;
    const int c = (int)(i % D);
    T* a = gi + r * 3 * D;
    T* b = gh + r * 3 * D;
    float ar = ld<T>(a + c), az = ld<T>(a + D + c), an = ld<T>(a + 2 * D + c);
    float br = ld<T>(b + c), bz = ld<T>(b + D + c), bn = ld<T>(b + 2 * D + c);
    if (b_ih) {
        ar += b_ih[c]; az += b_ih[D + c]; an += b_ih[2 * D + c];
        br += b_hh[c]; bz += b_hh[D + c]; bn += b_hh[2 * D + c];
        st<T>(a + c, ar); st<T>(a + D + c, az); st<T>(a + 2 * D + c, an);
        st<T>(b + c, br); st<T>(b + D + c, bz); st<T>(b + 2 * D + c, bn);
        // the backward recomputes the gates from the STORED values: use them here too (bf16 storage rounds)
        ar = ld<T>(a + c); az = ld<T>(a + D + c); an = ld<T>(a + 2 * D + c);
        br = ld<T>(b + c); bz = ld<T>(b + D + c); bn = ld<T>(b + 2 * D + c);
    }
    const float rg = sigm(ar + br);
    const float zg = sigm(az + bz);
    const float ng = tanhf(an + rg * bn);
    st<T>(hn + i, (1.f - zg) * ng + zg * ld<T>(h + i));
}
template <typename T>
__global__ void gru_bwd_kernel(const T* __restrict__ gi, const T* __restrict__ gh, const T* __restrict__ h,
                               const T* __restrict__ dhn, T* __restrict__ dgi, T* __restrict__ dgh,
                               T* __restrict__ dh, T* __restrict__ zero_out, int R, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * D) return;
    if (zero_out) st<T>(zero_out + i, 0.f);
    const int64_t r = i / D;
    const int c = (int)(i % D);
    const T* a = gi + r * 3 * D;
    const T* b = gh + r * 3 * D;
    const float hn_ = ld<T>(b + 2 * D + c);
    const float rg = sigm(ld<T>(a + c) + ld<T>(b + c));
    const float zg = sigm(ld<T>(a + D + c) + ld<T>(b + D + c));
    const float ng = tanhf(ld<T>(a + 2 * D + c) + rg * hn_);
    const float hv = ld<T>(h + i), g = ld<T>(dhn + i);
    const float dn = g * (1.f - zg), dz = g * (hv - ng);
    const float dpre_n = dn * (1.f - ng * ng);
    const float dr = dpre_n * hn_;
    const float dpre_r = dr * rg * (1.f - rg), dpre_z = dz * zg * (1.f - zg);
    st<T>(dgi + r * 3 * D + c, dpre_r);
    st<T>(dgi + r * 3 * D + D + c, dpre_z);
    st<T>(dgi + r * 3 * D + 2 * D + c, dpre_n);
    st<T>(dgh + r * 3 * D + c, dpre_r);
    st<T>(dgh + r * 3 * D + D + c, dpre_z);
    st<T>(dgh + r * 3 * D + 2 * D + c, dpre_n * rg);
    st<T>(dh + i, g * zg);
}

// ---- per-RoI max over cells ----------------------------------------------------------------------
// x [K, cells, C] -> y [K, C], arg [K, C] (first cell holding the maximum).
// Scalar form (any C): one thread per (k, c), serial over the cells.
template <typename T>
__global__ void cell_amax_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int32_t* __restrict__ arg, int K,
                                     int cells, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)K * C) return;
    const int64_t k = i / C;
    const int c = (int)(i % C);
    const T* p = x + k * cells * C + c;
    float m = ld<T>(p);
    int a = 0;
    for (int j = 1; j < cells; ++j) {
        const float v = ld<T>(p + (int64_t)j * C);
        if (v > m) { m = v; a = j; }
    }
    st<T>(y + i, m);
    arg[i] = a;
}
// bf16, C % 256 == 0: block = (RoI k, 256 channels); thread = (8 channels, one of 8 cell phases): 16-byte loads, 8 cell
// phases in flight, combined through LDS (ties -> smallest cell index, as the serial scan).  (The scalar form ran 196
// dependent 2-byte loads per thread: 89 us for 77 MB.)
__global__ __launch_bounds__(256) void cell_amax_fwd_vec_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                                int32_t* __restrict__ arg, int cells, int C) {
    __shared__ float sm[8][256];
    __shared__ int sa[8][256];
    const int k = blockIdx.y, cb = blockIdx.x * 256;
    const int cg = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const bf16_t* p = x + (int64_t)k * cells * C + cb + cg * 8;
    float m[8];
    int a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; a[e] = 0x7fffffff; }
#pragma unroll 5                                                      // (five cell rows in flight per thread)
    for (int j = ph; j < cells; j += 8) {
        const uint4 r = *reinterpret_cast<const uint4*>(p + (int64_t)j * C);
        const float v[8] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                            __uint_as_float(r.y & 0xffff0000u), __uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
                            __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (v[e] > m[e] || a[e] == 0x7fffffff) { m[e] = v[e]; a[e] = j; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[ph][cg * 8 + e] = m[e]; sa[ph][cg * 8 + e] = a[e]; }
    __syncthreads();
    const int c = threadIdx.x;
    float bm = sm[0][c];
    int ba = sa[0][c];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        const float vm = sm[q][c];
        const int va = sa[q][c];
        if (va != 0x7fffffff && (ba == 0x7fffffff || vm > bm || (vm == bm && va < ba))) { bm = vm; ba = va; }
    }
    y[(int64_t)k * C + cb + c] = f32_to_bf16(bm);
    arg[(int64_t)k * C + cb + c] = ba == 0x7fffffff ? 0 : ba;
}
template <typename T>
__global__ void cell_amax_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ arg, T* __restrict__ dx,
                                     int K, int cells, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)K * cells * C) return;
    const int c = (int)(i % C);
    const int j = (int)((i / C) % cells);
    const int64_t k = i / ((int64_t)C * cells);
    st<T>(dx + i, arg[k * C + c] == j ? ld<T>(dy + k * C + c) : 0.f);
}
// bf16, C % 8 == 0: 8 channels (one 16-byte store) per thread
__global__ __launch_bounds__(256) void cell_amax_bwd_vec_kernel(const bf16_t* __restrict__ dy, const int32_t* __restrict__ arg,
                                                                bf16_t* __restrict__ dx, int64_t n8, int cells, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c8 = C >> 3;
    const int cg = (int)(i % c8);
    const int j = (int)((i / c8) % cells);
    const int64_t k = i / ((int64_t)c8 * cells);
    const int4 a0 = *reinterpret_cast<const int4*>(arg + k * C + cg * 8);
    const int4 a1 = *reinterpret_cast<const int4*>(arg + k * C + cg * 8 + 4);
    const uint4 g = *reinterpret_cast<const uint4*>(dy + k * C + cg * 8);
    uint4 o;
    o.x = (a0.x == j ? (g.x & 0x0000ffffu) : 0u) | (a0.y == j ? (g.x & 0xffff0000u) : 0u);
    o.y = (a0.z == j ? (g.y & 0x0000ffffu) : 0u) | (a0.w == j ? (g.y & 0xffff0000u) : 0u);
    o.z = (a1.x == j ? (g.z & 0x0000ffffu) : 0u) | (a1.y == j ? (g.z & 0xffff0000u) : 0u);
    o.w = (a1.z == j ? (g.w & 0x0000ffffu) : 0u) | (a1.w == j ? (g.w & 0xffff0000u) : 0u);
    *reinterpret_cast<uint4*>(dx + i * 8) = o;
}

inline unsigned nblk(int64_t n, int per) { return (unsigned)cdiv64(n, per); }

}  // namespace

#define DISPATCH_T(dtype, EXPR_T)                       \
    do {                                                \
        if ((dtype) == FOCUS_BF16) { typedef bf16_t T; EXPR_T; } \
        else { typedef float T; EXPR_T; }               \
    } while (0)

extern "C" int focus_colsum(const void* x, float* out, int M, int N, int64_t row_stride, int accumulate, int dtype,
                            void* stream) {
    if (!x || !out) return FOCUS_ERR_NULL;
    if (M <= 0 || N <= 0) return FOCUS_OK;
    hipStream_t s = (hipStream_t)stream;
    // a single row strip stores its sums directly: no zero fill, no atomics (short matrices: slot / object-token layers)
    const bool vec = dtype == FOCUS_BF16 && (N & 7) == 0 && (row_stride & 7) == 0 && focus_aligned(x, 16);
    // small problems (the narrow layers of reduced test models, stacked over the applications of a loop): ONE strip however
    // long, i.e. a fixed summation order -- with several strips the fp32 atomics below make the sums differ in the last bit
    // from run to run, which a bit-exact replay test cannot tell from a real divergence.  Large ones keep the parallel strips.
    const bool small = (int64_t)M * N <= (4 << 20);
    const bool one_strip = small || M <= (vec ? 512 : 256);
    if (!accumulate && !one_strip && hipMemsetAsync(out, 0, sizeof(float) * N, s) != hipSuccess) return FOCUS_ERR_LAUNCH;
    if (vec) {
        const int rpb = one_strip ? M : 512;
        hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, s,
                           (const bf16_t*)x, out, M, N, row_stride, rpb, accumulate);
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    const int rpb = one_strip ? M : 256;
    dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, s, (const T*)x, out, M, N,
                                         row_stride, rpb, accumulate));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

// one 64x64 tile of one tensor per block: fp32 in, bf16 row-major out and bf16 transposed out through LDS
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const focus_shadow_item* __restrict__ items) {
    __shared__ bf16_t tile[64][68];
    const focus_shadow_item it = items[blockIdx.y];
    const int tiles_c = (it.cols + 63) >> 6, tiles_r = (it.rows + 63) >> 6;
    if ((int)blockIdx.x >= tiles_r * tiles_c) return;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const int t = threadIdx.x, lr = t >> 4, lc = (t & 15) * 4;
    bf16_t* dst = static_cast<bf16_t*>(it.dst);
    bf16_t* dstT = static_cast<bf16_t*>(it.dstT);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = tr * 64 + lr + 16 * k, c = tc * 64 + lc;
        uint2 o = make_uint2(0, 0);
        if (r < it.rows && c < it.cols) {
            const float4 v = *reinterpret_cast<const float4*>(it.src + (int64_t)r * it.cols + c);
            o.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
            o.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
            if (dst) *reinterpret_cast<uint2*>(dst + (int64_t)r * it.cols + c) = o;
        }
        *reinterpret_cast<uint2*>(&tile[lr + 16 * k][lc]) = o;
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int oc = tc * 64 + lr + 16 * k, orr = tr * 64 + lc;      // output row = source column
        if (oc < it.cols && orr < it.rows) {
            uint2 o;
            o.x = (uint32_t)tile[lc + 0][lr + 16 * k] | ((uint32_t)tile[lc + 1][lr + 16 * k] << 16);
            o.y = (uint32_t)tile[lc + 2][lr + 16 * k] | ((uint32_t)tile[lc + 3][lr + 16 * k] << 16);
            *reinterpret_cast<uint2*>(dstT + (int64_t)oc * it.rows + orr) = o;
        }
    }
}

extern "C" int focus_shadow_refresh(const focus_shadow_item* items, int n_items, int max_rows, int max_cols, void* stream) {
    if (!items) return FOCUS_ERR_NULL;
    if (n_items <= 0) return FOCUS_OK;
    if (n_items > 65535 || max_rows <= 0 || max_cols <= 0) return FOCUS_ERR_SHAPE;
    const int tiles = ((max_rows + 63) / 64) * ((max_cols + 63) / 64);
    hipLaunchKernelGGL(shadow_refresh_kernel, dim3(tiles, n_items), dim3(256), 0, (hipStream_t)stream, items);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    if (!src || !dst) return FOCUS_ERR_NULL;
    if (n <= 0) return FOCUS_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t sa = sd == FOCUS_BF16 ? 8 : 16, da = dd == FOCUS_BF16 ? 8 : 16;
    if (!focus_aligned(src, sa) || !focus_aligned(dst, da)) return FOCUS_ERR_ALIGN;
    unsigned g = (unsigned)std::min<int64_t>(cdiv64(n, 1024), 8192);
    if (sd == FOCUS_F32 && dd == FOCUS_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (sd == FOCUS_BF16 && dd == FOCUS_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (sd == FOCUS_F32 && dd == FOCUS_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, n);
    else
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_transpose_pad(const void* src, int sd, int64_t src_ld, int64_t src_bs, void* dst, int dd,
                                   int64_t dst_ld, int64_t dst_bs, int R, int Cc, int batch, void* stream) {
    if (!src || !dst) return FOCUS_ERR_NULL;
    if (R <= 0 || Cc <= 0 || batch <= 0) return FOCUS_OK;
    if (dst_ld < R || batch > 65535) return FOCUS_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((Cc + 63) / 64, (unsigned)cdiv64(dst_ld, 64), batch);
    if (grid.y > 65535) return FOCUS_ERR_SHAPE;
#define TP(TS, TD) hipLaunchKernelGGL((transpose_pad_kernel<TS, TD>), grid, dim3(256), 0, s, (const TS*)src, src_ld, src_bs, (TD*)dst, dst_ld, dst_bs, R, Cc)
    if (sd == FOCUS_F32 && dd == FOCUS_F32) TP(float, float);
    else if (sd == FOCUS_F32 && dd == FOCUS_BF16) TP(float, bf16_t);
    else if (sd == FOCUS_BF16 && dd == FOCUS_BF16) TP(bf16_t, bf16_t);
    else TP(bf16_t, float);
#undef TP
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_diag_gather(const void* xt, void* xdiag, int B, int S, int F, int C, int dtype, void* stream) {
    if (!xt || !xdiag) return FOCUS_ERR_NULL;
    if ((C & 3) || F <= 0 || S % F) return FOCUS_ERR_SHAPE;
    const int64_t rows = (int64_t)B * S;
    DISPATCH_T(dtype, hipLaunchKernelGGL((diag_gather_kernel<T>), dim3(nblk(rows * C / 4, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)xt, (T*)xdiag, rows, S, F, S / F, C));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
extern "C" int focus_diag_scatter_add(const void* dxdiag, void* dxt, int B, int S, int F, int C, int dtype,
                                      void* stream) {
    if (!dxdiag || !dxt) return FOCUS_ERR_NULL;
    if ((C & 3) || F <= 0 || S % F) return FOCUS_ERR_SHAPE;
    const int64_t rows = (int64_t)B * S;
    DISPATCH_T(dtype, hipLaunchKernelGGL((diag_scatter_add_kernel<T>), dim3(nblk(rows * C / 4, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)dxdiag, (T*)dxt, rows, S, F, S / F, C));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_im2col_patches(const float* x, void* cols, int B, int Cin, int Tn, int H, int W, int kt, int kh,
                                    int kw, int dtype, void* stream) {
    if (!x || !cols) return FOCUS_ERR_NULL;
    if (kt <= 0 || kh <= 0 || kw <= 0 || Tn % kt || H % kh || W % kw || (kw & 3) || (W & 3)) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(x, 16)) return FOCUS_ERR_ALIGN;
    const int64_t total = (int64_t)B * (Tn / kt) * (H / kh) * (W / kw) * Cin * kt * kh * kw / 4;
    DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_kernel<T>), dim3(nblk(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, x, (T*)cols, B, Cin, Tn, H, W, kt, kh, kw));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_embed_assemble(const void* patch, const float* cls, const float* pos, const float* temp,
                                    void* tokens, int B, int Tn, int P, int C, int dtype, void* stream) {
    if (!patch || !cls || !pos || !temp || !tokens) return FOCUS_ERR_NULL;
    if (C & 3) return FOCUS_ERR_SHAPE;
    const int64_t total = (int64_t)B * (1 + (int64_t)Tn * P) * C / 4;
    DISPATCH_T(dtype, hipLaunchKernelGGL((embed_assemble_kernel<T>), dim3(nblk(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)patch, cls, pos, temp, (T*)tokens, B, Tn, P, C));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_xent_ls(const float* logits, const int64_t* target, float* loss_rows, float* dlogits, int R,
                             int Ncls, float smoothing, void* stream) {
    if (!logits || !target || !loss_rows || !dlogits) return FOCUS_ERR_NULL;
    if (R <= 0 || Ncls <= 0) return FOCUS_ERR_SHAPE;
    hipLaunchKernelGGL(xent_ls_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, target, loss_rows, dlogits,
                       R, Ncls, smoothing);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_gru_gates_fwd(void* gi, void* gh, const void* h, void* hn, const float* b_ih, const float* b_hh, int R,
                                   int D, int dtype, void* stream) {
    if (!gi || !gh || !h || !hn || (!b_ih != !b_hh)) return FOCUS_ERR_NULL;
    DISPATCH_T(dtype, hipLaunchKernelGGL((gru_fwd_kernel<T>), dim3(nblk((int64_t)R * D, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (T*)gi, (T*)gh, (const T*)h, (T*)hn, b_ih, b_hh, R, D));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
extern "C" int focus_gru_gates_bwd(const void* gi, const void* gh, const void* h, const void* dhn, void* dgi,
                                   void* dgh, void* dh, void* zero_out, int R, int D, int dtype, void* stream) {
    if (!gi || !gh || !h || !dhn || !dgi || !dgh || !dh) return FOCUS_ERR_NULL;
    DISPATCH_T(dtype, hipLaunchKernelGGL((gru_bwd_kernel<T>), dim3(nblk((int64_t)R * D, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)gi, (const T*)gh, (const T*)h, (const T*)dhn,
                                         (T*)dgi, (T*)dgh, (T*)dh, (T*)zero_out, R, D));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

extern "C" int focus_cell_amax_fwd(const void* x, void* y, int32_t* arg, int K, int cells, int C, int dtype,
                                   void* stream) {
    if (!x || !y || !arg) return FOCUS_ERR_NULL;
    if (cells <= 0) return FOCUS_ERR_SHAPE;
    if (K <= 0) return FOCUS_OK;
    if (dtype == FOCUS_BF16 && (C & 255) == 0 && K <= 65535 && focus_aligned(x, 16)) {
        hipLaunchKernelGGL(cell_amax_fwd_vec_kernel, dim3(C / 256, K), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (bf16_t*)y, arg, cells, C);
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((cell_amax_fwd_kernel<T>), dim3(nblk((int64_t)K * C, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, (T*)y, arg, K, cells, C));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}
extern "C" int focus_cell_amax_bwd(const void* dy, const int32_t* arg, void* dx, int K, int cells, int C, int dtype,
                                   void* stream) {
    if (!dy || !arg || !dx) return FOCUS_ERR_NULL;
    if (K <= 0 || cells <= 0) return FOCUS_OK;
    if (dtype == FOCUS_BF16 && (C & 7) == 0 && focus_aligned(dy, 16) && focus_aligned(dx, 16) && focus_aligned(arg, 16)) {
        const int64_t n8 = (int64_t)K * cells * (C >> 3);
        hipLaunchKernelGGL(cell_amax_bwd_vec_kernel, dim3(nblk(n8, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dy, arg, (bf16_t*)dx, n8, cells, C);
        FOCUS_CHECK_LAUNCH();
        return FOCUS_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((cell_amax_bwd_kernel<T>), dim3(nblk((int64_t)K * cells * C, 256)),
                                         dim3(256), 0, (hipStream_t)stream, (const T*)dy, arg, (T*)dx, K, cells, C));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

template <typename T>
__global__ void scale_add_kernel(const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ scale,
                                 float keep, T* __restrict__ out, int64_t total, int64_t per) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;       // 16-byte pieces (per % 8 == 0)
    if (i >= total) return;
    float sc = scale[i / per];
    if (keep > 0.f) sc = __fdiv_rn(floorf(__fadd_rn(keep, sc)), keep);   // scale = the uniform draw: mask / keep_prob
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[8];
    ld8<T>(y + i, b);
    if (x) ld8<T>(x + i, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += sc * b[e];
    st8<T>(out + i, a);
}
extern "C" int focus_scale_add(const void* x, const void* y, const float* scale, float keep, void* out, int B,
                               int64_t per, int dtype, void* stream) {
    if (!y || !scale || !out) return FOCUS_ERR_NULL;
    if (B <= 0 || per <= 0) return FOCUS_OK;
    if (per & 7) return FOCUS_ERR_SHAPE;
    if (!focus_aligned(y, 16) || !focus_aligned(out, 16) || (x && !focus_aligned(x, 16))) return FOCUS_ERR_ALIGN;
    const int64_t total = (int64_t)B * per;
    DISPATCH_T(dtype, hipLaunchKernelGGL((scale_add_kernel<T>), dim3(nblk(total / 8, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, (const T*)y, scale, keep, (T*)out, total, per));
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void tr16_probe_kernel(int16_t* out) {
    __shared__ __attribute__((aligned(16))) int16_t img[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) img[i] = (int16_t)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
    auto* ptr = (__attribute__((address_space(3))) s16x4*)(img + (4 * g + q) * 64 + 4 * pp);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}
extern "C" int focus_debug_tr16_probe(int16_t* out, void* stream) {
    if (!out) return FOCUS_ERR_NULL;
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    FOCUS_CHECK_LAUNCH();
    return FOCUS_OK;
}

static const char* kErr[] = {"ok", "bad shape", "unsupported dtype", "misaligned pointer or stride",
                             "HIP launch failure", "null pointer", "workspace too small"};
extern "C" const char* focus_strerror(int status) {
    const int i = -status;
    return (i >= 0 && i <= 6) ? kErr[i] : "unknown focus status";
}
extern "C" int focus_abi_version(void) { return 2; }
