// Write-pattern probe (developer tool, not part of the library): how fast can 16-byte stores fill an (M x N) bf16 matrix when a
// workgroup owns (a) whole rows, (b) a 128-column tile of 128 rows (the tiled kernels' epilogue), (c) a 64-column tile?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/wprobe tools/probe/wprobe.hip ; tools/probe/wprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// tile writer: workgroup (256 threads) writes rows [128 bm, +128) x columns [TW bn, +TW) with TW*2/16 lanes per row piece
template <int TW>
__global__ __launch_bounds__(256) void k_tile(uint16_t* out, int64_t M, int N, int tiles_n) {
    const int bn = blockIdx.x % tiles_n;
    const int64_t bm = blockIdx.x / tiles_n;
    constexpr int LPR = TW * 2 / 16;      // lanes per row piece
    constexpr int RPI = 256 / LPR;        // rows per workgroup store sweep
    const int r0 = threadIdx.x / LPR, c = threadIdx.x % LPR;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll 4
    for (int r = r0; r < 128; r += RPI) {
        const int64_t m = bm * 128 + r;
        if (m < M) *reinterpret_cast<u32x4*>(out + m * N + bn * TW + c * 8) = v;
    }
}
// row writer: workgroup writes 32 whole rows (persistent-free: one block per 32 rows)
__global__ __launch_bounds__(256) void k_rows(uint16_t* out, int64_t M, int N) {
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    const int vec_per_row = N / 8;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int i = threadIdx.x; i < 32 * vec_per_row; i += 256) {
        const int r = i / vec_per_row, c = i - r * vec_per_row;
        if (m0 + r < M) *reinterpret_cast<u32x4*>(out + (m0 + r) * N + c * 8) = v;
    }
}
// copy-like reader+writer: read (M x K) once, write (M x N) as tiles (K = 96: the fc1 / qkv forward shape)
template <int TW>
__global__ __launch_bounds__(256) void k_tile_rw(const uint16_t* in, uint16_t* out, int64_t M, int K, int N, int tiles_n) {
    const int bn = blockIdx.x % tiles_n;
    const int64_t bm = blockIdx.x / tiles_n;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < 128 * K / 8; i += 256) {
        const int r = i / (K / 8), c = i - r * (K / 8);
        const int64_t m = bm * 128 + r;
        if (m < M) acc ^= *reinterpret_cast<const u32x4*>(in + m * K + c * 8);
    }
    constexpr int LPR = TW * 2 / 16;
    constexpr int RPI = 256 / LPR;
    const int r0 = threadIdx.x / LPR, c = threadIdx.x % LPR;
#pragma unroll 4
    for (int r = r0; r < 128; r += RPI) {
        const int64_t m = bm * 128 + r;
        if (m < M) *reinterpret_cast<u32x4*>(out + m * N + bn * TW + c * 8) = acc;
    }
}

template <typename F>
static float timeit(F f, int n = 10) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / n;
}

int main() {
    const int64_t M = 401408;
    uint16_t *buf, *in;
    hipMalloc(&buf, (size_t)M * 1536 * 2 * 2);
    hipMalloc(&in, (size_t)M * 96 * 2);
    hipMemset(in, 1, (size_t)M * 96 * 2);
    for (int N : {128, 384, 768, 1536}) {
        const double gb = (double)M * N * 2 / 1e9;
        float ms = timeit([&] { hipLaunchKernelGGL(k_rows, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, 0, buf, M, N); });
        printf("N=%4d rows      : %7.1f us  %5.2f TB/s\n", N, ms * 1e3, gb / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_tile<128>, dim3((unsigned)((M + 127) / 128 * (N / 128))), dim3(256), 0, 0, buf, M, N, N / 128); });
        printf("N=%4d tile128   : %7.1f us  %5.2f TB/s\n", N, ms * 1e3, gb / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_tile<64>, dim3((unsigned)((M + 127) / 128 * (N / 64))), dim3(256), 0, 0, buf, M, N, N / 64); });
        printf("N=%4d tile64    : %7.1f us  %5.2f TB/s\n", N, ms * 1e3, gb / ms);
        const double gbrw = gb + (double)M * 96 * 2 / 1e9 ;
        ms = timeit([&] { hipLaunchKernelGGL(k_tile_rw<128>, dim3((unsigned)((M + 127) / 128 * (N / 128))), dim3(256), 0, 0, in, buf, M, 96, N, N / 128); });
        printf("N=%4d tile128 rw: %7.1f us  %5.2f TB/s (incl. one read of X per n-tile from L2)\n", N, ms * 1e3, gbrw / ms);
    }
    // five outputs of N = 384 written by one workgroup per tile (the T = 4 forward): 5 separate matrices
    {
        const int N = 384;
        const double gb = 5.0 * M * N * 2 / 1e9;
        float ms = timeit([&] {
            for (int o = 0; o < 5; ++o)
                hipLaunchKernelGGL(k_tile<128>, dim3((unsigned)((M + 127) / 128 * (N / 128))), dim3(256), 0, 0, buf + (size_t)o * M * N, M, N, N / 128);
        });
        printf("5 x N=384 tile128 (5 launches): %7.1f us  %5.2f TB/s\n", ms * 1e3, gb / ms);
    }
    return 0;
}
