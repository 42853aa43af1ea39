// glue.hip -- per-token glue around the MTLoRA hot path, HBM-bound streaming kernels.
//
//   k_ln_fwd / k_ln_bwd   LayerNorm over the last dim, fp32 or bf16 input, fp32 or bf16 OUTPUT written directly
//                         in the dtype the following MTLoRALinear consumes (the reference's autocast path
//                         writes an fp32 normalised tensor and then casts it: 2.5x the bytes), fp32 statistics.
//                         A row is handled by LPR lanes (8..64) holding it entirely in registers (two-pass mean /
//                         variance, no E[x^2] cancellation); 64/LPR rows per wave-instruction, 16 B per lane.
//                         Backward also produces dgamma / dbeta: per-thread column accumulators over the rows a
//                         workgroup visits, LDS reduction over its row groups, per-workgroup partials, and a
//                         deterministic second-stage reduce.
#include "common.h"

namespace {

constexpr int LN_MAXV = 8;  // 16-byte vectors per lane per row (C <= 64 lanes * 8 vec * VEC)

template <typename T>
__device__ __forceinline__ void ld_vec(const T* p, float (&f)[8]);
template <>
__device__ __forceinline__ void ld_vec<float>(const float* p, float (&f)[8]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0];
    f[1] = v[1];
    f[2] = v[2];
    f[3] = v[3];
}
template <>
__device__ __forceinline__ void ld_vec<bf16>(const bf16* p, float (&f)[8]) {
    Vec16<bf16> v = mtl_ld16<bf16>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v.e[e];
}
// store NE consecutive elements
template <typename T, int NE>
__device__ __forceinline__ void st_vec(T* p, const float* f) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < NE; e += 4) *reinterpret_cast<f32x4*>(p + e) = f32x4{f[e], f[e + 1], f[e + 2], f[e + 3]};
    } else if constexpr (NE == 8) {
        Vec16<bf16> v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = (bf16)f[e];
        *reinterpret_cast<u32x4*>(p) = v.raw;
    } else {
        bf16x4 v = {(bf16)f[0], (bf16)f[1], (bf16)f[2], (bf16)f[3]};
        *reinterpret_cast<bf16x4*>(p) = v;
    }
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct LnParams {
    const void* x;
    const float* gamma;
    const float* beta;
    void* y;
    float* mean;
    float* rstd;
    // backward
    const void* dy;
    void* dx;
    float* part;  // [gridDim.x][2][C]
    int64_t M;
    int C;
    float eps;
};

template <typename TI, typename TO, int LPR>
__global__ __launch_bounds__(256) void k_ln_fwd(const LnParams p) {
    constexpr int VE = ET<TI>::VEC;
    constexpr int RPW = 64 / LPR;  // rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, lr = lane % LPR;
    const int nvec = p.C / VE;
    const TI* x = reinterpret_cast<const TI*>(p.x);
    TO* y = reinterpret_cast<TO*>(p.y);
    float g[LN_MAXV][VE], b[LN_MAXV][VE];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int v = lr + i * LPR;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            g[i][e] = v < nvec ? p.gamma[v * VE + e] : 0.f;
            b[i][e] = v < nvec ? p.beta[v * VE + e] : 0.f;
        }
    }
    const int64_t rows_per_blk = 4 * RPW;
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_blk; r0 < p.M; r0 += (int64_t)gridDim.x * rows_per_blk) {
        const int64_t row = r0 + wave * RPW + sub;
        const bool rv = row < p.M;
        float f[LN_MAXV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int v = lr + i * LPR;
            if (rv && v < nvec) {
                ld_vec<TI>(x + row * p.C + v * VE, f[i]);
#pragma unroll
                for (int e = 0; e < VE; ++e) s += f[i][e];
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) f[i][e] = 0.f;
            }
        }
        const float mean = group_sum<LPR>(s) / p.C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int v = lr + i * LPR;
            if (v < nvec) {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float d = f[i][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(group_sum<LPR>(q) / p.C + p.eps);
        if (rv) {
            if (lr == 0) {
                p.mean[row] = mean;
                p.rstd[row] = rstd;
            }
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int v = lr + i * LPR;
                if (v < nvec) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < VE; ++e) o[e] = (f[i][e] - mean) * rstd * g[i][e] + b[i][e];
                    st_vec<TO, VE>(y + row * p.C + v * VE, o);
                }
            }
        }
    }
}

// TX: dtype of x and dx; TG: dtype of dy
template <typename TX, typename TG, int LPR>
__global__ __launch_bounds__(256) void k_ln_bwd(const LnParams p) {
    constexpr int VE = ET<TX>::VEC;  // elements per lane-vector (x drives the vector width; dy read with the same count)
    constexpr int RPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 * RPW row groups][2][C] column partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, lr = lane % LPR;
    const int nvec = p.C / VE;
    const TX* x = reinterpret_cast<const TX*>(p.x);
    const TG* dy = reinterpret_cast<const TG*>(p.dy);
    TX* dx = reinterpret_cast<TX*>(p.dx);
    float g[LN_MAXV][VE], ag[LN_MAXV][VE], ab[LN_MAXV][VE];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int v = lr + i * LPR;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            g[i][e] = v < nvec ? p.gamma[v * VE + e] : 0.f;
            ag[i][e] = 0.f;
            ab[i][e] = 0.f;
        }
    }
    const int64_t rows_per_blk = 4 * RPW;
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_blk; r0 < p.M; r0 += (int64_t)gridDim.x * rows_per_blk) {
        const int64_t row = r0 + wave * RPW + sub;
        const bool rv = row < p.M;
        const float mean = rv ? p.mean[row] : 0.f, rstd = rv ? p.rstd[row] : 0.f;
        float xh[LN_MAXV][VE], gy[LN_MAXV][VE];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int v = lr + i * LPR;
            if (rv && v < nvec) {
                float fx[8], fg[8];
                ld_vec<TX>(x + row * p.C + v * VE, fx);
                if constexpr (sizeof(TG) == sizeof(TX)) {
                    ld_vec<TG>(dy + row * p.C + v * VE, fg);
                } else if constexpr (sizeof(TG) == 2) {  // x fp32 (4 per vec), dy bf16: 4 elements = 8 bytes
                    bf16x4 t = *reinterpret_cast<const bf16x4*>(dy + row * p.C + v * VE);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fg[e] = (float)t[e];
                } else {  // x bf16 (8 per vec), dy fp32: two 16-byte loads
                    f32x4 t0 = *reinterpret_cast<const f32x4*>(dy + row * p.C + v * VE);
                    f32x4 t1 = *reinterpret_cast<const f32x4*>(dy + row * p.C + v * VE + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fg[e] = t0[e];
                        fg[4 + e] = t1[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float h = (fx[e] - mean) * rstd;
                    xh[i][e] = h;
                    ag[i][e] += fg[e] * h;
                    ab[i][e] += fg[e];
                    const float t = fg[e] * g[i][e];
                    gy[i][e] = t;
                    c1 += t;
                    c2 += t * h;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    xh[i][e] = 0.f;
                    gy[i][e] = 0.f;
                }
            }
        }
        c1 = group_sum<LPR>(c1) / p.C;
        c2 = group_sum<LPR>(c2) / p.C;
        if (rv) {
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int v = lr + i * LPR;
                if (v < nvec) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < VE; ++e) o[e] = rstd * (gy[i][e] - c1 - xh[i][e] * c2);
                    st_vec<TX, VE>(dx + row * p.C + v * VE, o);
                }
            }
        }
    }
    // workgroup reduction of the column accumulators: the 4 waves x RPW row groups hold the same columns; each
    // group writes its own LDS slab and the slabs are summed in a fixed order (deterministic, no float atomics)
    float* mine = sm + (size_t)(wave * RPW + sub) * 2 * p.C;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int v = lr + i * LPR;
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                mine[v * VE + e] = ag[i][e];
                mine[p.C + v * VE + e] = ab[i][e];
            }
        }
    }
    __syncthreads();
    float* dst = p.part + (int64_t)blockIdx.x * 2 * p.C;
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
        float t = 0.f;
        for (int gi = 0; gi < 4 * RPW; ++gi) t += sm[(size_t)gi * 2 * p.C + i];
        dst[i] = t;
    }
}

// second stage: one workgroup per 64 columns; its 4 waves stride over the partial rows (coalesced 256-byte
// reads), then a fixed-order LDS combine -> deterministic
__global__ __launch_bounds__(256) void k_ln_reduce(const float* part, float* dgamma, float* dbeta, int nblk, int C) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // column in [0, 2C)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < 2 * C) {
        int b = wave;
        for (; b + 12 < nblk; b += 16) {
            a0 += part[(int64_t)b * 2 * C + c];
            a1 += part[(int64_t)(b + 4) * 2 * C + c];
            a2 += part[(int64_t)(b + 8) * 2 * C + c];
            a3 += part[(int64_t)(b + 12) * 2 * C + c];
        }
        for (; b < nblk; b += 4) a0 += part[(int64_t)b * 2 * C + c];
    }
    sm[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && c < 2 * C) {
        const float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        if (c < C)
            dgamma[c] = t;
        else
            dbeta[c - C] = t;
    }
}

int pick_lpr(int nvec) {
    int lpr = 8;
    while (lpr < 64 && (nvec + lpr - 1) / lpr > 3) lpr *= 2;
    return lpr;
}

int ln_grid(int64_t M, int lpr) {
    const int64_t rows_per_blk = 4 * (64 / lpr);
    int64_t g = mtl_ceil_div(M, rows_per_blk);
    if (g > 256 * 2) g = 256 * 2;  // also the number of dgamma/dbeta partials the second stage sums
    return (int)(g < 1 ? 1 : g);
}

int ln_check(int64_t M, int64_t C, int xdt, int ydt) {
    if ((xdt != MTLORA_F32 && xdt != MTLORA_BF16) || (ydt != MTLORA_F32 && ydt != MTLORA_BF16)) return MTLORA_ERR_DTYPE;
    const int ve = xdt == MTLORA_F32 ? 4 : 8;
    if (M < 0 || C <= 0 || C % ve) return MTLORA_ERR_SHAPE;
    if (mtl_ceil_div(C / ve, 64) > LN_MAXV) return MTLORA_ERR_UNSUPPORTED;
    return MTLORA_OK;
}

}  // namespace

extern "C" {

int64_t mtlora_layernorm_bwd_scratch_bytes(int64_t M, int64_t C, int x_dtype) {
    if (ln_check(M, C, x_dtype, MTLORA_F32) != MTLORA_OK) return -1;
    const int lpr = pick_lpr((int)(C / (x_dtype == MTLORA_F32 ? 4 : 8)));
    return (int64_t)ln_grid(M, lpr) * 2 * C * 4 + 256;
}

#define LN_DISPATCH_LPR(KERNEL, ...)                                                          \
    switch (lpr) {                                                                            \
        case 8: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 8>), dim3(grid), dim3(256), lds, s, p); break;   \
        case 16: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 16>), dim3(grid), dim3(256), lds, s, p); break; \
        case 32: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 32>), dim3(grid), dim3(256), lds, s, p); break; \
        default: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 64>), dim3(grid), dim3(256), lds, s, p); break; \
    }

int mtlora_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t M, int64_t C, float eps, int x_dtype, int y_dtype, void* stream) {
    int st = ln_check(M, C, x_dtype, y_dtype);
    if (st != MTLORA_OK) return st;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)y) & 15u) return MTLORA_ERR_ALIGN;
    if (M == 0) return MTLORA_OK;
    LnParams p = {};
    p.x = x;
    p.gamma = gamma;
    p.beta = beta;
    p.y = y;
    p.mean = mean;
    p.rstd = rstd;
    p.M = M;
    p.C = (int)C;
    p.eps = eps;
    const int lpr = pick_lpr((int)(C / (x_dtype == MTLORA_F32 ? 4 : 8)));
    const int grid = ln_grid(M, lpr);
    const size_t lds = 0;
    hipStream_t s = (hipStream_t)stream;
    const int es_x = mtl_elem_size(x_dtype), es_y = mtl_elem_size(y_dtype);
    MtlProfScope prof(PK_LN_FWD, (double)M * C * (es_x + es_y), s);
    if (x_dtype == MTLORA_F32 && y_dtype == MTLORA_F32) {
        LN_DISPATCH_LPR(k_ln_fwd, float, float)
    } else if (x_dtype == MTLORA_F32) {
        LN_DISPATCH_LPR(k_ln_fwd, float, bf16)
    } else if (y_dtype == MTLORA_F32) {
        LN_DISPATCH_LPR(k_ln_fwd, bf16, float)
    } else {
        LN_DISPATCH_LPR(k_ln_fwd, bf16, bf16)
    }
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         void* dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int x_dtype, int dy_dtype,
                         void* scratch, int64_t scratch_bytes, void* stream) {
    int st = ln_check(M, C, x_dtype, dy_dtype);
    if (st != MTLORA_OK) return st;
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !scratch) return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)scratch) & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_layernorm_bwd_scratch_bytes(M, C, x_dtype) - 256) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        hipMemsetAsync(dgamma, 0, (size_t)C * 4, s);
        hipMemsetAsync(dbeta, 0, (size_t)C * 4, s);
        return MTLORA_OK;
    }
    LnParams p = {};
    p.x = x;
    p.dy = dy;
    p.gamma = gamma;
    p.mean = const_cast<float*>(mean);
    p.rstd = const_cast<float*>(rstd);
    p.dx = dx;
    p.part = reinterpret_cast<float*>(scratch);
    p.M = M;
    p.C = (int)C;
    const int lpr = pick_lpr((int)(C / (x_dtype == MTLORA_F32 ? 4 : 8)));
    const int grid = ln_grid(M, lpr);
    const size_t lds = (size_t)4 * (64 / lpr) * 2 * C * 4;
    const int es_x = mtl_elem_size(x_dtype), es_g = mtl_elem_size(dy_dtype);
    {
        MtlProfScope prof(PK_LN_BWD, (double)M * C * (2 * es_x + es_g), s);
        if (x_dtype == MTLORA_F32 && dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, float, float)
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, float, bf16)
        } else if (dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_bwd, bf16, bf16)
        }
    }
    hipLaunchKernelGGL(k_ln_reduce, dim3((unsigned)mtl_ceil_div(2 * C, 64)), dim3(256), 0, s, (const float*)p.part, dgamma,
                       dbeta, grid, (int)C);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}
}
