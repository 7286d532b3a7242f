// Micro-probe: per-CU throughput of 16-byte-per-lane global loads for two lane->address patterns over the same bytes.
//   pattern 0 (MFMA fragment): lane (r = l & 15, g = l >> 4) reads row r, bytes 16 g .. 16 g + 15 of a 64-byte k-step:
//                              16 rows x 64 B per instruction (what the temporal kernels issue)
//   pattern 1 (row contiguous): lane l reads row l >> 3, bytes 16 (l & 7) ..: 8 rows x 128 B per instruction
// Rows are `pitch` bytes apart (1536 = the x~ / q2 row).  One workgroup of 512 threads per CU, each wave walks tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void probe(const char* __restrict__ base, int64_t pitch, int tiles, int pattern, float* sink, int resident) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc = 0.f;
    // every block walks its own region so that the loads are L2/HBM traffic, not L1 hits
    const char* p0 = base + (resident ? (int64_t)(blockIdx.x & 7) * 8 * 128 * pitch : (int64_t)blockIdx.x * tiles * 128 * pitch);
    for (int t = 0; t < tiles; ++t) {
        const char* tile = p0 + (int64_t)(resident ? (t & 7) : t) * 128 * pitch;   // resident: 8 tiles (1.5 MB span) per XCD, L2 hits
#pragma unroll
        for (int i = 0; i < 2; ++i) {                              // 2 instructions cover 16 rows x 128 B per wave either way
            const char* a;
            if (pattern == 0) a = tile + (int64_t)(16 * w + (lane & 15)) * pitch + 64 * i + 16 * (lane >> 4);
            else a = tile + (int64_t)(16 * w + 8 * i + (lane >> 3)) * pitch + 16 * (lane & 7);
            const float4 v = *reinterpret_cast<const float4*>(a);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int64_t pitch = 1536;
    const int tiles = 64, blocks = 256;
    const size_t bytes = (size_t)blocks * tiles * 128 * pitch;
    char* d; float* sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 4);
    hipMemset(d, 0, bytes);
    for (int resident = 0; resident < 2; ++resident)
    for (int pattern = 0; pattern < 2; ++pattern) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d, pitch, tiles, pattern, sink, resident);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double useful = (double)blocks * tiles * 128 * 128;   // bytes actually loaded
            if (rep == 2) printf("resident %d pattern %d: %.1f us, %.2f TB/s useful, %.0f ns per tile per CU\n", resident, pattern, ms * 1e3, useful / ms / 1e9, ms * 1e6 / tiles);
        }
    }
    return 0;
}
