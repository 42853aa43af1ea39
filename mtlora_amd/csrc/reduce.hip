// reduce.hip -- small deterministic reductions of the train step's callers that ATen would run as multi-block
// reduce_kernels: those zero a semaphore buffer with hipMemsetAsync first, and on this ROCm stack a CAPTURED small memset stops
// taking effect from the second graph replay on (mtl_harness.GraphedTrainStep) -- the nine memsets per step listed by
// tools/find_memsets.py were exactly these sums.  Two launches each (per-block partials in a fixed order, then one combine
// kernel): no atomics, no memset, bit-reproducible.
//
//   mtlora_colsum       out[n] = sum_m x[m][n]                     bias gradient of the heads' big-M 1x1 convolutions
//                                                                  (dL/db of reference models/seg_hrnet.py:498-526 layers)
//   mtlora_label_stat   kind 0: #{ lab != ignore }                 valid-pixel count / mask sum of mtl_loss_schemes.py:22-39,
//                       kind 1: mean(1 - (lab >= 0.5))             :162-220; class balance w of :42-89
#include "common.h"

namespace {

constexpr int RD_MAXBLK = 1024;

// stage 1: block b sums rows b, b + G, ... (256 / nvec rows per sweep) of the 16-byte column vectors -> part[b][N]
template <typename T>
__global__ __launch_bounds__(256) void k_colsum_part(const T* x, int64_t M, int N, float* part) {
    constexpr int VE = ET<T>::VEC;
    __shared__ float sm[256][VE + 1];
    const int nvec = N / VE;
    const int tid = threadIdx.x;
    for (int v0 = 0; v0 < nvec; v0 += 256) {  // column-vector chunks (one pass for N <= 256 * VE)
        const int cv = nvec - v0 < 256 ? nvec - v0 : 256;  // vectors in this chunk
        const int rps = 256 / cv;                          // rows per sweep
        const int r = tid / cv, v = tid - r * cv;
        float acc[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = 0.f;
        if (r < rps) {
            for (int64_t m = (int64_t)blockIdx.x * rps + r; m < M; m += (int64_t)gridDim.x * rps) {
                const Vec16<T> q = mtl_ld16<T>(x + m * N + (int64_t)(v0 + v) * VE);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[e] += mtl_to_f32(q.e[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) sm[tid][e] = (r < rps) ? acc[e] : 0.f;
        __syncthreads();
        if (tid < cv) {  // fixed-order combine over the rows of the sweep
            float t[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) t[e] = 0.f;
            for (int rr = 0; rr < rps; ++rr)
#pragma unroll
                for (int e = 0; e < VE; ++e) t[e] += sm[rr * cv + tid][e];
#pragma unroll
            for (int e = 0; e < VE; ++e) part[(int64_t)blockIdx.x * N + (int64_t)(v0 + tid) * VE + e] = t[e];
        }
        __syncthreads();
    }
}

// stage 2: out[c] = scale * sum_b part[b][c]  (16 waves share the partials round-robin, 4 independent chains each)
constexpr int RD_RW = 16;
__global__ __launch_bounds__(64 * RD_RW) void k_colsum_combine(const float* part, float* out, int nblk, int N, float scale) {
    __shared__ float sm[RD_RW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < N) {
        int b = wave;
        for (; b + 3 * RD_RW < nblk; b += 4 * RD_RW) {
            a0 += part[(int64_t)b * N + c];
            a1 += part[(int64_t)(b + RD_RW) * N + c];
            a2 += part[(int64_t)(b + 2 * RD_RW) * N + c];
            a3 += part[(int64_t)(b + 3 * RD_RW) * N + c];
        }
        for (; b < nblk; b += RD_RW) a0 += part[(int64_t)b * N + c];
    }
    sm[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < RD_RW; ++w) t += sm[w][lane];
        out[c] = t * scale;
    }
}

// label statistics, stage 1: per-block count (exact in fp32: a block sees < 2^24 elements)
__global__ __launch_bounds__(256) void k_label_stat_part(const float* lab, int64_t n, int kind, float ignore, float* part) {
    __shared__ float sm[256];
    float c = 0.f;
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lab + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) c += kind == 0 ? (v[e] != ignore ? 1.f : 0.f) : (v[e] >= 0.5f ? 0.f : 1.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {  // tail elements
        const float v = lab[(nv << 2) + threadIdx.x];
        c += kind == 0 ? (v != ignore ? 1.f : 0.f) : (v >= 0.5f ? 0.f : 1.f);
    }
    sm[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

// label statistics, stage 2: the per-block counts are exact integers < 2^24; their total may exceed 2^24, so it is formed in
// double and rounded once (what ATen's int64 sum followed by .float() gives)
__global__ __launch_bounds__(256) void k_label_stat_combine(const float* part, float* out, int nblk, double scale) {
    __shared__ double sm[256];
    double t = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) t += (double)part[b];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sm[0] * scale);
}

int rd_blocks(int64_t work_items) {
    int64_t g = mtl_ceil_div(work_items, 256 * 8);
    if (g > RD_MAXBLK) g = RD_MAXBLK;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int64_t mtlora_colsum_scratch_bytes(int64_t M, int64_t N) {
    if (M < 0 || N <= 0) return -1;
    return (int64_t)RD_MAXBLK * N * 4 + 256;
}

int mtlora_colsum(const void* x, int64_t M, int64_t N, int dtype, float* out, void* scratch, int64_t scratch_bytes, void* stream) {
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16) return MTLORA_ERR_DTYPE;
    const int ve = dtype == MTLORA_F32 ? 4 : 8;
    if (M < 0 || N <= 0 || N % ve || N > (1 << 20)) return MTLORA_ERR_SHAPE;
    if (!x || !out || !scratch) return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)scratch) & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_colsum_scratch_bytes(M, N)) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nvec = (int)(N / ve);
    const int rps = nvec >= 256 ? 1 : 256 / nvec;
    int64_t g = mtl_ceil_div(M, (int64_t)rps * 8);  // >= 8 sweeps per block
    if (g > RD_MAXBLK) g = RD_MAXBLK;
    if (g < 1) g = 1;
    float* part = reinterpret_cast<float*>(scratch);
    if (dtype == MTLORA_F32)
        hipLaunchKernelGGL(k_colsum_part<float>, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const float*>(x), M, (int)N, part);
    else
        hipLaunchKernelGGL(k_colsum_part<bf16>, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const bf16*>(x), M, (int)N, part);
    hipLaunchKernelGGL(k_colsum_combine, dim3((unsigned)mtl_ceil_div(N, 64)), dim3(64 * RD_RW), 0, s, (const float*)part, out, (int)g,
                       (int)N, 1.f);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int64_t mtlora_label_stat_scratch_bytes(int64_t n) {
    if (n < 0) return -1;
    return (int64_t)RD_MAXBLK * 4 + 256;
}

int mtlora_label_stat(const float* label, int64_t n, int kind, float ignore_index, float* out, void* scratch,
                      int64_t scratch_bytes, void* stream) {
    if (kind != 0 && kind != 1) return MTLORA_ERR_UNSUPPORTED;
    if (n <= 0) return MTLORA_ERR_SHAPE;
    if (!label || !out || !scratch) return MTLORA_ERR_NULL;
    if ((uintptr_t)label & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_label_stat_scratch_bytes(n)) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int g = rd_blocks(n / 4 + 1);
    float* part = reinterpret_cast<float*>(scratch);
    hipLaunchKernelGGL(k_label_stat_part, dim3((unsigned)g), dim3(256), 0, s, label, n, kind, ignore_index, part);
    hipLaunchKernelGGL(k_label_stat_combine, dim3(1), dim3(256), 0, s, (const float*)part, out, g, kind == 0 ? 1.0 : 1.0 / (double)n);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

}  // extern "C"
