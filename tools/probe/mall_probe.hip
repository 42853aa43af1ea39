// Infinity-Cache (256 MB, memory side) re-use probe (developer tool, not part of the library).
// Question: when kernel B streams a tensor that kernel A has just streamed (written or read) front to back, does B find the
// tensor's TAIL in the Infinity Cache -- i.e. is B faster when it walks the tensor back to front?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mall_probe tools/probe/mall_probe.hip ; tools/probe/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int TPB = 256, UNR = 4;
constexpr int64_t BLK_VEC = (int64_t)TPB * UNR;  // 16-byte vectors per workgroup step (16 KB)

__global__ __launch_bounds__(TPB) void k_write(u32x4* buf, int64_t nblk, int rev) {
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const int64_t bb = rev ? nblk - 1 - b : b;
        u32x4* p = buf + bb * BLK_VEC + threadIdx.x;
#pragma unroll
        for (int u = 0; u < UNR; ++u) p[u * TPB] = v;
    }
}
__global__ __launch_bounds__(TPB) void k_read(const u32x4* buf, int64_t nblk, int rev, u32x4* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const int64_t bb = rev ? nblk - 1 - b : b;
        const u32x4* p = buf + bb * BLK_VEC + threadIdx.x;
        u32x4 t[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) t[u] = p[u * TPB];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= t[u];
    }
    if (acc[0] == 0x12345678u) sink[threadIdx.x] = acc;
}
__global__ __launch_bounds__(TPB) void k_copy(const u32x4* src, u32x4* dst, int64_t nblk, int rev) {
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const int64_t bb = rev ? nblk - 1 - b : b;
        const u32x4* p = src + bb * BLK_VEC + threadIdx.x;
        u32x4* q = dst + bb * BLK_VEC + threadIdx.x;
        u32x4 t[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) t[u] = p[u * TPB];
#pragma unroll
        for (int u = 0; u < UNR; ++u) q[u * TPB] = t[u];
    }
}

int main() {
    const int grid = 256 * 8;
    const int64_t MB = 1 << 20;
    u32x4 *a, *b, *c, *flush, *sink;
    const int64_t maxb = 768 * MB;
    hipMalloc(&a, maxb);
    hipMalloc(&b, maxb);
    hipMalloc(&c, maxb);
    hipMalloc(&flush, 1024 * MB);
    hipMalloc(&sink, 4096);
    hipMemset(a, 1, maxb);
    hipMemset(b, 1, maxb);
    hipMemset(c, 1, maxb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto do_flush = [&]() { k_write<<<grid, TPB>>>(flush, 1024 * MB / (BLK_VEC * 16), 0); };
    const int sizes[] = {32, 64, 96, 128, 192, 256, 320, 384, 512, 768};
    printf("%6s | %9s %9s %9s | %9s %9s | %9s %9s | %9s %9s   (GB/s of the SECOND kernel's own bytes)\n", "MB", "R cold", "W cold", "C cold", "W>R fwd", "W>R rev",
           "R>R fwd", "R>R rev", "C>C fwd", "C>C rev");
    for (int sz : sizes) {
        const int64_t bytes = sz * MB, nblk = bytes / (BLK_VEC * 16);
        auto timed = [&](auto first, auto second, double moved) {
            float best = 1e30f;
            for (int it = 0; it < 5; ++it) {
                do_flush();
                first();
                hipEventRecord(e0);
                second();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            return moved / best * 1e-6;
        };
        auto none = [&]() {};
        auto rd = [&](int rev) { return [=]() { k_read<<<grid, TPB>>>(a, nblk, rev, sink); }; };
        auto wr = [&](int rev) { return [=]() { k_write<<<grid, TPB>>>(a, nblk, rev); }; };
        auto cp1 = [&](int rev) { return [=]() { k_copy<<<grid, TPB>>>(a, b, nblk, rev); }; };
        auto cp2 = [&](int rev) { return [=]() { k_copy<<<grid, TPB>>>(b, c, nblk, rev); }; };
        const double r_cold = timed(none, rd(0), (double)bytes), w_cold = timed(none, wr(0), (double)bytes), c_cold = timed(none, cp1(0), 2.0 * bytes);
        const double wr_f = timed(wr(0), rd(0), (double)bytes), wr_r = timed(wr(0), rd(1), (double)bytes);
        const double rr_f = timed(rd(0), rd(0), (double)bytes), rr_r = timed(rd(0), rd(1), (double)bytes);
        const double cc_f = timed(cp1(0), cp2(0), 2.0 * bytes), cc_r = timed(cp1(0), cp2(1), 2.0 * bytes);
        printf("%6d | %9.0f %9.0f %9.0f | %9.0f %9.0f | %9.0f %9.0f | %9.0f %9.0f\n", sz, r_cold, w_cold, c_cold, wr_f, wr_r, rr_f, rr_r, cc_f, cc_r);
    }
    return 0;
}
