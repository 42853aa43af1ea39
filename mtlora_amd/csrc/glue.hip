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
#include "internal.h"

namespace {

constexpr int LN_MAXV = 8;  // most 16-byte vectors per lane per row (C <= 64 lanes * 8 vec * VEC); kernels are specialised on 3 / 8

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
template <>
__device__ __forceinline__ void ld_vec<f16>(const f16* p, float (&f)[8]) {
    Vec16<f16> v = mtl_ld16<f16>(p);
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
        Vec16<T> v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = (T)f[e];
        *reinterpret_cast<u32x4*>(p) = v.raw;
    } else {
        *reinterpret_cast<u32x2*>(p) = u32x2{mtl_pack2<T>(f[0], f[1]), mtl_pack2<T>(f[2], f[3])};
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
    const void* add;  // optional (dtype of x): dx = add + LayerNorm-backward(dy) -- the gradient of the skip path that forks off x
    float* part;  // [gridDim.x][2][C]
    int64_t M;
    int C;
    float eps;
    // patch-merging gather (PatchMerging: 2x2 neighbourhood concat before its LayerNorm): x / dx / add are (B, mg_H*mg_W, C/4)
    // token tensors and row r = (b, y2, x2) of the normalised (M, C) matrix is [x(2y2,2x2) | x(2y2+1,2x2) | x(2y2,2x2+1) |
    // x(2y2+1,2x2+1)]; mg_W == 0: plain rows
    int mg_H, mg_W;
    // fused residual (Swin block: x_new = shortcut + DropPath(branch) immediately followed by LayerNorm(x_new)):
    //   forward : x = shortcut, rb = branch (dtype of y), xsum = x_new (dtype of x) written by the kernel, then normalised
    //   backward: dbr = rscale[sample] * dx (dtype of dy): the branch gradient, written next to dx (= the shortcut gradient)
    const void* rb;
    void* xsum;
    void* dbr;
    const float* rscale;  // [B] per-sample DropPath scale or null (1)
    int64_t rows_per_sample;
    // multi-stream form (task-enabled block: ONE shortcut, 1+T branches, 1+T normalised outputs): stream k = blockIdx.y of the
    // forward launch / an inner loop of k_resln_bwd_multi; rscale is then [nk][B]
    int nk;
    const void* rb_k[MTLORA_MAX_TASKS + 1];
    void* xsum_k[MTLORA_MAX_TASKS + 1];
    void* y_k[MTLORA_MAX_TASKS + 1];
    float* mean_k[MTLORA_MAX_TASKS + 1];
    float* rstd_k[MTLORA_MAX_TASKS + 1];
    const void* dy_k[MTLORA_MAX_TASKS + 1];
    const void* add_k[MTLORA_MAX_TASKS + 1];
    void* dbr_k[MTLORA_MAX_TASKS + 1];
    // independent streams through the SAME LayerNorm in one launch (multi_x: blockIdx.y selects x / y / statistics, backward
    // dy / dx / addend and a partial-sum slab; PatchMerging's norm over the shared + task tensors)
    int multi_x;
    const void* x_k[MTLORA_MAX_TASKS + 1];
    void* dx_k[MTLORA_MAX_TASKS + 1];
};
// arrays of the parameter block are indexed through the kernarg segment (constant address space): dynamic indexing of the
// by-value copy would move the whole struct to scratch
typedef const __attribute__((address_space(4))) LnParams* LnKargs;

// row / d for 0 <= row < rows, 0 < d <= rows.  The int64 quotient costs ~100 VALU instructions per lane (software division),
// once per row group -- more than the arithmetic of a 96-wide row; every shape in use has rows < 2^31, where one 32-bit unsigned
// division (~20 instructions) gives the same result.  `rows` is wave-uniform.
__device__ __forceinline__ int64_t row_div(int64_t row, int64_t d, int64_t rows) {
    if (rows <= (int64_t)0x7FFFFFFF) return (int64_t)((uint32_t)row / (uint32_t)d);
    return row / d;
}

// element offset of column `col` (a multiple of the vector width) of row `row` in x / dx
struct LnRow {
    int64_t base;  // plain: row * C; merged: offset of token (2y2, 2x2)
};
__device__ __forceinline__ LnRow ln_row(const LnParams& p, int64_t row) {
    LnRow r;
    if (p.mg_W == 0) {
        r.base = row * p.C;
    } else {
        const int W2 = p.mg_W >> 1, H2 = p.mg_H >> 1, Cs = p.C >> 2;
        const int64_t b = row_div(row, (int64_t)H2 * W2, p.M);
        const int rem = (int)(row - b * H2 * W2);
        const int y2 = rem / W2, x2 = rem - y2 * W2;
        r.base = ((b * p.mg_H + 2 * y2) * p.mg_W + 2 * x2) * (int64_t)Cs;
    }
    return r;
}
// per-lane constant part: offset of column `col` relative to the row base
__device__ __forceinline__ int64_t ln_col(const LnParams& p, int col) {
    if (p.mg_W == 0) return col;
    const int Cs = p.C >> 2, q = col / Cs, within = col - q * Cs;
    return ((int64_t)(q & 1) * p.mg_W + (q >> 1)) * Cs + within;  // (dy = q & 1, dx = q >> 1)
}

// raw 16-byte vector -> floats
template <typename T>
__device__ __forceinline__ void cvt_vec(const u32x4& r, float (&f)[8]);
template <>
__device__ __forceinline__ void cvt_vec<float>(const u32x4& r, float (&f)[8]) {
    const f32x4 v = __builtin_bit_cast(f32x4, r);
    f[0] = v[0];
    f[1] = v[1];
    f[2] = v[2];
    f[3] = v[3];
}
template <>
__device__ __forceinline__ void cvt_vec<bf16>(const u32x4& r, float (&f)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f[2 * q] = __builtin_bit_cast(float, r[q] << 16);
        f[2 * q + 1] = __builtin_bit_cast(float, r[q] & 0xFFFF0000u);
    }
}
template <>
__device__ __forceinline__ void cvt_vec<f16>(const u32x4& r, float (&f)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f[2 * q] = mtl_lo2<f16>(r[q]);
        f[2 * q + 1] = mtl_hi2<f16>(r[q]);
    }
}

// UNR row groups per wave iteration: their loads are issued back to back and kept as raw 16-byte vectors (a wave with a
// single 1.5 KB row group in flight per iteration ran at 1.5-2.8 TB/s for the stage-1..3 shapes)
// NE elements of type T -> floats (NE = 4 or 8)
template <typename T, int NE>
__device__ __forceinline__ void ld_n(const T* p, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < NE; e += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + e);
            f[e] = v[0];
            f[e + 1] = v[1];
            f[e + 2] = v[2];
            f[e + 3] = v[3];
        }
    } else if constexpr (NE == 8) {
        cvt_vec<T>(*reinterpret_cast<const u32x4*>(p), f);
    } else {
        const u32x2 r = *reinterpret_cast<const u32x2*>(p);
        f[0] = mtl_lo2<T>(r[0]);
        f[1] = mtl_hi2<T>(r[0]);
        f[2] = mtl_lo2<T>(r[1]);
        f[3] = mtl_hi2<T>(r[1]);
    }
}
// floats -> one raw 16-byte vector of T (4 fp32 or 8 bf16)
template <typename T>
__device__ __forceinline__ u32x4 pack_vec(const float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(u32x4, f32x4{f[0], f[1], f[2], f[3]});
    } else {
        Vec16<T> v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = (T)f[e];
        return v.raw;
    }
}

template <typename TI, typename TO, int LPR, int MAXV, bool RES>
__global__ __launch_bounds__(256) void k_ln_fwd(const LnParams p) {
    constexpr int VE = ET<TI>::VEC;
    constexpr int RPW = 64 / LPR;  // rows per wave per group
    constexpr int UNR = MAXV <= 3 ? 4 : 1;  // (the wide-row specialisation already holds 8 vectors per lane)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, lr = lane % LPR;
    const int nvec = p.C / VE;
    const TI* x = reinterpret_cast<const TI*>(p.x);
    TO* y = reinterpret_cast<TO*>(p.y);
    float* mean_out = p.mean;
    float* rstd_out = p.rstd;
    const void* rb_ptr = p.rb;
    void* xs_ptr = p.xsum;
    const float* rscale = p.rscale;
    if (p.multi_x) {
        LnKargs K = (LnKargs)__builtin_amdgcn_kernarg_segment_ptr();
        const int k = blockIdx.y;
        x = reinterpret_cast<const TI*>(K->x_k[k]);
        y = reinterpret_cast<TO*>(K->y_k[k]);
        mean_out = K->mean_k[k];
        rstd_out = K->rstd_k[k];
        if constexpr (RES) {  // independent streams, each with its own residual: x_k + s_k * branch_k
            rb_ptr = K->rb_k[k];
            xs_ptr = K->xsum_k[k];
            if (rscale) rscale += (int64_t)k * (p.M / p.rows_per_sample);
        }
    }
    if constexpr (RES) {
        if (p.nk > 0) {  // multi-stream launch: blockIdx.y selects the branch / outputs; the shortcut x is shared (an in-kernel
                         // loop over the streams that reads it once was slower: 473 vs 424 us at stage 0 -- fewer workgroups)
            LnKargs K = (LnKargs)__builtin_amdgcn_kernarg_segment_ptr();
            const int k = blockIdx.y;
            y = reinterpret_cast<TO*>(K->y_k[k]);
            mean_out = K->mean_k[k];
            rstd_out = K->rstd_k[k];
            rb_ptr = K->rb_k[k];
            xs_ptr = K->xsum_k[k];
            if (rscale) rscale += (int64_t)k * (p.M / p.rows_per_sample);
        }
    }
    float g[MAXV][VE], b[MAXV][VE];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lr + i * LPR;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            g[i][e] = v < nvec ? p.gamma[v * VE + e] : 0.f;
            b[i][e] = v < nvec ? p.beta[v * VE + e] : 0.f;
        }
    }
    int64_t coff[MAXV];  // x offset of this lane's vectors relative to the row base (plain or patch-merging gather)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) coff[i] = ln_col(p, (lr + i * LPR < nvec ? lr + i * LPR : 0) * VE);
    const float inv_c = 1.f / (float)p.C;
    const int64_t rows_per_blk = 4 * RPW * UNR;
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_blk; r0 < p.M; r0 += (int64_t)gridDim.x * rows_per_blk) {
        u32x4 raw[UNR][MAXV];
        int64_t row[UNR], xbs[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            row[u] = r0 + (int64_t)(wave * UNR + u) * RPW + sub;
            const int64_t xb = ln_row(p, row[u] < p.M ? row[u] : 0).base;
            xbs[u] = xb;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lr + i * LPR;
                raw[u][i] = (row[u] < p.M && v < nvec) ? *reinterpret_cast<const u32x4*>(x + xb + coff[i])
                                                      : u32x4{0u, 0u, 0u, 0u};
            }
        }
        if constexpr (RES) {  // x_new = shortcut + s * branch, stored (rounded to the stream dtype) and normalised
            const TO* rb = reinterpret_cast<const TO*>(rb_ptr);
            TI* xs = reinterpret_cast<TI*>(xs_ptr);
            // the branch vectors and the per-sample scales are fetched for ALL row groups before the first x_new store: the
            // compiler cannot move a load above a store that may alias it, so loading inside the store loop left one branch
            // vector in flight at a time (the shortcut loads above are already issued back to back)
            constexpr bool PRE = sizeof(TO) * VE <= 16;  // a branch vector fits one 16-byte register (all but fp32 branch / bf16 x)
            u32x4 braw[PRE ? UNR : 1][PRE ? MAXV : 1];
            float scv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool rok = row[u] < p.M;
                scv[u] = (rok && rscale) ? rscale[row_div(row[u], p.rows_per_sample, p.M)] : 1.f;
                if constexpr (PRE) {
#pragma unroll
                    for (int i = 0; i < MAXV; ++i) {
                        const int v = lr + i * LPR;
                        u32x4 b = {0u, 0u, 0u, 0u};
                        if (rok && v < nvec) {
                            if constexpr (sizeof(TO) * VE == 16) {
                                b = *reinterpret_cast<const u32x4*>(rb + xbs[u] + coff[i]);
                            } else {
                                const u32x2 h2 = *reinterpret_cast<const u32x2*>(rb + xbs[u] + coff[i]);
                                b[0] = h2[0];
                                b[1] = h2[1];
                            }
                        }
                        braw[u][i] = b;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (row[u] >= p.M) continue;
                const float sc = scv[u];
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lr + i * LPR;
                    if (v < nvec) {
                        float fb[8], fx[8];
                        // (branch and x_new share the layout of x: plain rows, or the token tensor of the merge gather)
                        if constexpr (!PRE) {
                            ld_n<TO, VE>(rb + xbs[u] + coff[i], fb);
                        } else if constexpr (sizeof(TO) == 4) {
                            const f32x4 q4 = __builtin_bit_cast(f32x4, braw[u][i]);
                            fb[0] = q4[0];
                            fb[1] = q4[1];
                            fb[2] = q4[2];
                            fb[3] = q4[3];
                        } else if constexpr (VE == 8) {
                            cvt_vec<TO>(braw[u][i], fb);
                        } else {
                            fb[0] = mtl_lo2<TO>(braw[u][i][0]);
                            fb[1] = mtl_hi2<TO>(braw[u][i][0]);
                            fb[2] = mtl_lo2<TO>(braw[u][i][1]);
                            fb[3] = mtl_hi2<TO>(braw[u][i][1]);
                        }
                        cvt_vec<TI>(raw[u][i], fx);
#pragma unroll
                        for (int e = 0; e < VE; ++e) fx[e] += sc * fb[e];
                        raw[u][i] = pack_vec<TI>(fx);
                        *reinterpret_cast<u32x4*>(xs + xbs[u] + coff[i]) = raw[u][i];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float f[8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                cvt_vec<TI>(raw[u][i], f);
#pragma unroll
                for (int e = 0; e < VE; ++e) s += f[e];  // out-of-range vectors are zero
            }
            const float mean = group_sum<LPR>(s) * inv_c;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (lr + i * LPR < nvec) {
                    cvt_vec<TI>(raw[u][i], f);
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        const float d = f[e] - mean;
                        q += d * d;
                    }
                }
            }
            const float rstd = rsqrtf(group_sum<LPR>(q) * inv_c + p.eps);
            if (row[u] < p.M) {
                if (lr == 0) {
                    mean_out[row[u]] = mean;
                    rstd_out[row[u]] = rstd;
                }
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lr + i * LPR;
                    if (v < nvec) {
                        cvt_vec<TI>(raw[u][i], f);
                        float o[8];
#pragma unroll
                        for (int e = 0; e < VE; ++e) o[e] = (f[e] - mean) * rstd * g[i][e] + b[i][e];
                        st_vec<TO, VE>(y + row[u] * p.C + v * VE, o);
                    }
                }
            }
        }
    }
}

// TX: dtype of x and dx; TG: dtype of dy
template <typename TX, typename TG, int LPR, int MAXV>
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
    const TX* addp = reinterpret_cast<const TX*>(p.add);
    const float* mean_in = p.mean;
    const float* rstd_in = p.rstd;
    void* dbr_ptr = p.dbr;
    const float* rscale_in = p.rscale;
    float* part_out = p.part + (int64_t)blockIdx.x * 2 * p.C;
    if (p.multi_x) {
        LnKargs K = (LnKargs)__builtin_amdgcn_kernarg_segment_ptr();
        const int k = blockIdx.y;
        x = reinterpret_cast<const TX*>(K->x_k[k]);
        dy = reinterpret_cast<const TG*>(K->dy_k[k]);
        dx = reinterpret_cast<TX*>(K->dx_k[k]);
        addp = reinterpret_cast<const TX*>(K->add_k[k]);
        mean_in = K->mean_k[k];
        rstd_in = K->rstd_k[k];
        part_out = p.part + ((int64_t)k * gridDim.x + blockIdx.x) * 2 * p.C;
        dbr_ptr = K->dbr_k[k];
        if (rscale_in) rscale_in += (int64_t)k * (p.M / p.rows_per_sample);
    }
    float g[MAXV][VE], ag[MAXV][VE], ab[MAXV][VE];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lr + i * LPR;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            g[i][e] = v < nvec ? p.gamma[v * VE + e] : 0.f;
            ag[i][e] = 0.f;
            ab[i][e] = 0.f;
        }
    }
    int64_t coff[MAXV];  // x / dx / addend offsets of this lane's vectors relative to the row base
#pragma unroll
    for (int i = 0; i < MAXV; ++i) coff[i] = ln_col(p, (lr + i * LPR < nvec ? lr + i * LPR : 0) * VE);
    // UNR row groups per wave iteration, loads issued back to back as raw vectors before any arithmetic (as in k_ln_fwd)
    // (bf16 rows hold 8 elements per vector: two row groups in flight need > 256 VGPRs -- one wave per SIMD -- and ran slower)
    constexpr int UNR = (MAXV <= 3 && sizeof(TX) == 4) ? 2 : 1;
    constexpr int GW = sizeof(TG) == sizeof(TX) ? 4 : (sizeof(TG) == 2 ? 2 : 8);  // dwords of dy per lane-vector
    const int64_t rows_per_blk = 4 * RPW * UNR;
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_blk; r0 < p.M; r0 += (int64_t)gridDim.x * rows_per_blk) {
        u32x4 rx[UNR][MAXV], ra[UNR][MAXV];
        uint32_t rg[UNR][MAXV][GW];
        int64_t row[UNR], xb[UNR];
        float mean[UNR], rstd[UNR], scv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            row[u] = r0 + (int64_t)(wave * UNR + u) * RPW + sub;
            const bool rv = row[u] < p.M;
            xb[u] = ln_row(p, rv ? row[u] : 0).base;
            mean[u] = rv ? mean_in[row[u]] : 0.f;
            rstd[u] = rv ? rstd_in[row[u]] : 0.f;
            // (fetched with the other operands: a load placed after the dx stores cannot be hoisted above them)
            scv[u] = (rv && dbr_ptr && rscale_in) ? rscale_in[row_div(row[u], p.rows_per_sample, p.M)] : 1.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lr + i * LPR;
                const bool ok = rv && v < nvec;
                rx[u][i] = ok ? *reinterpret_cast<const u32x4*>(x + xb[u] + coff[i]) : u32x4{0u, 0u, 0u, 0u};
                const TG* gp = dy + (rv ? row[u] : 0) * p.C + (v < nvec ? v : 0) * VE;
                if constexpr (GW == 4) {
                    const u32x4 t = ok ? *reinterpret_cast<const u32x4*>(gp) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int q = 0; q < 4; ++q) rg[u][i][q] = t[q];
                } else if constexpr (GW == 2) {
                    const u32x2 t = ok ? *reinterpret_cast<const u32x2*>(gp) : u32x2{0u, 0u};
                    rg[u][i][0] = t[0];
                    rg[u][i][1] = t[1];
                } else {
                    const u32x4 t0 = ok ? *reinterpret_cast<const u32x4*>(gp) : u32x4{0u, 0u, 0u, 0u};
                    const u32x4 t1 = ok ? *reinterpret_cast<const u32x4*>(gp + 4) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rg[u][i][q] = t0[q];
                        rg[u][i][4 + q] = t1[q];
                    }
                }
                ra[u][i] = (ok && addp) ? *reinterpret_cast<const u32x4*>(addp + xb[u] + coff[i]) : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bool rv = row[u] < p.M;
            float xh[MAXV][VE], gy[MAXV][VE];
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                float fx[8], fg[8];
                cvt_vec<TX>(rx[u][i], fx);
                if constexpr (GW == 4) {
                    cvt_vec<TG>(u32x4{rg[u][i][0], rg[u][i][1], rg[u][i][2], rg[u][i][3]}, fg);
                } else if constexpr (GW == 2) {  // x fp32 (4 per vector), dy bf16
                    fg[0] = mtl_lo2<TG>(rg[u][i][0]);
                    fg[1] = mtl_hi2<TG>(rg[u][i][0]);
                    fg[2] = mtl_lo2<TG>(rg[u][i][1]);
                    fg[3] = mtl_hi2<TG>(rg[u][i][1]);
                } else {  // x bf16 (8 per vector), dy fp32
#pragma unroll
                    for (int q = 0; q < 8; ++q) fg[q] = __builtin_bit_cast(float, rg[u][i][q]);
                }
                // out-of-range rows / vectors were loaded as zeros with mean = rstd = 0: every term below is then 0
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float h = (fx[e] - mean[u]) * rstd[u];
                    xh[i][e] = h;
                    ag[i][e] += fg[e] * h;
                    ab[i][e] += fg[e];
                    const float t = fg[e] * g[i][e];
                    gy[i][e] = t;
                    c1 += t;
                    c2 += t * h;
                }
            }
            c1 = group_sum<LPR>(c1) / p.C;
            c2 = group_sum<LPR>(c2) / p.C;
            if (rv) {
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lr + i * LPR;
                    if (v < nvec) {
                        float o[8], fa[8];
                        cvt_vec<TX>(ra[u][i], fa);
#pragma unroll
                        for (int e = 0; e < VE; ++e) o[e] = rstd[u] * (gy[i][e] - c1 - xh[i][e] * c2) + fa[e];
                        st_vec<TX, VE>(dx + xb[u] + coff[i], o);
                        if (dbr_ptr) {  // gradient of the residual branch (layout of x / dx): DropPath scale, dtype of dy
                            const float sc = scv[u];
#pragma unroll
                            for (int e = 0; e < VE; ++e) o[e] *= sc;
                            st_vec<TG, VE>(reinterpret_cast<TG*>(dbr_ptr) + xb[u] + coff[i], o);
                        }
                    }
                }
            }
        }
    }
    // workgroup reduction of the column accumulators: the 4 waves x RPW row groups hold the same columns; each
    // group writes its own LDS slab and the slabs are summed in a fixed order (deterministic, no float atomics)
    float* mine = sm + (size_t)(wave * RPW + sub) * 2 * p.C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
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
    float* dst = part_out;
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
        float t = 0.f;
        for (int gi = 0; gi < 4 * RPW; ++gi) t += sm[(size_t)gi * 2 * p.C + i];
        dst[i] = t;
    }
}

// Backward of the multi-stream residual + LayerNorm (task-enabled block half): for its rows a wave walks the nk streams --
// two in flight -- and forms per stream  dx_k = add_k + LN'(dy_k)  (never stored),  d_branch_k = s_k dx_k  (stored, dtype of
// dy), while  d_shortcut = sum_k dx_k  and the dgamma / dbeta partials accumulate in registers across the streams: one pass
// instead of nk LayerNorm backward launches + their reduces + the shared-residual backward (which re-read all nk dx_k).
template <typename TX, typename TG, int LPR, int MAXV>
__global__ __launch_bounds__(256) void k_resln_bwd_multi(const LnParams p) {
    constexpr int VE = ET<TX>::VEC;
    constexpr int RPW = 64 / LPR;
    constexpr int GW = sizeof(TG) == sizeof(TX) ? 4 : (sizeof(TG) == 2 ? 2 : 8);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    LnKargs K = (LnKargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, lr = lane % LPR;
    const int nvec = p.C / VE;
    TX* dsh = reinterpret_cast<TX*>(p.dx);
    const int64_t Bn = p.M / p.rows_per_sample;
    float g[MAXV][VE], ag[MAXV][VE], ab[MAXV][VE];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lr + i * LPR;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            g[i][e] = v < nvec ? p.gamma[v * VE + e] : 0.f;
            ag[i][e] = 0.f;
            ab[i][e] = 0.f;
        }
    }
    struct Regs {
        u32x4 rx[MAXV], ra[MAXV];
        uint32_t rg[MAXV][GW];
        float mean, rstd, sc;
    };
    const int64_t rows_per_blk = 4 * RPW;
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_blk; r0 < p.M; r0 += (int64_t)gridDim.x * rows_per_blk) {
        const int64_t row = r0 + wave * RPW + sub;
        const bool rv = row < p.M;
        const int64_t rbase = (rv ? row : 0) * p.C;
        float dsum[MAXV][VE];
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int e = 0; e < VE; ++e) dsum[i][e] = 0.f;
        auto load = [&](Regs& R, int k) __attribute__((always_inline)) {
            const TX* x = reinterpret_cast<const TX*>(K->xsum_k[k]);
            const TG* dy = reinterpret_cast<const TG*>(K->dy_k[k]);
            const TX* addp = reinterpret_cast<const TX*>(K->add_k[k]);
            R.mean = rv ? K->mean_k[k][row] : 0.f;
            R.rstd = rv ? K->rstd_k[k][row] : 0.f;
            R.sc = (rv && p.rscale) ? p.rscale[(int64_t)k * Bn + row_div(row, p.rows_per_sample, p.M)] : 1.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lr + i * LPR;
                const bool ok = rv && v < nvec;
                const int64_t off = rbase + (v < nvec ? v : 0) * VE;
                R.rx[i] = ok ? *reinterpret_cast<const u32x4*>(x + off) : u32x4{0u, 0u, 0u, 0u};
                R.ra[i] = (ok && addp) ? *reinterpret_cast<const u32x4*>(addp + off) : u32x4{0u, 0u, 0u, 0u};
                const TG* gp = dy + off;
                if constexpr (GW == 4) {
                    const u32x4 t = ok ? *reinterpret_cast<const u32x4*>(gp) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int q = 0; q < 4; ++q) R.rg[i][q] = t[q];
                } else if constexpr (GW == 2) {
                    const u32x2 t = ok ? *reinterpret_cast<const u32x2*>(gp) : u32x2{0u, 0u};
                    R.rg[i][0] = t[0];
                    R.rg[i][1] = t[1];
                } else {
                    const u32x4 t0 = ok ? *reinterpret_cast<const u32x4*>(gp) : u32x4{0u, 0u, 0u, 0u};
                    const u32x4 t1 = ok ? *reinterpret_cast<const u32x4*>(gp + 4) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        R.rg[i][q] = t0[q];
                        R.rg[i][4 + q] = t1[q];
                    }
                }
            }
        };
        auto compute = [&](Regs& R, int k) __attribute__((always_inline)) {
            TG* dbr = reinterpret_cast<TG*>(K->dbr_k[k]);
            float xh[MAXV][VE], gy[MAXV][VE];
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                float fx[8], fg[8];
                cvt_vec<TX>(R.rx[i], fx);
                if constexpr (GW == 4) {
                    cvt_vec<TG>(u32x4{R.rg[i][0], R.rg[i][1], R.rg[i][2], R.rg[i][3]}, fg);
                } else if constexpr (GW == 2) {
                    fg[0] = mtl_lo2<TG>(R.rg[i][0]);
                    fg[1] = mtl_hi2<TG>(R.rg[i][0]);
                    fg[2] = mtl_lo2<TG>(R.rg[i][1]);
                    fg[3] = mtl_hi2<TG>(R.rg[i][1]);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) fg[q] = __builtin_bit_cast(float, R.rg[i][q]);
                }
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float h = (fx[e] - R.mean) * R.rstd;
                    xh[i][e] = h;
                    ag[i][e] += fg[e] * h;
                    ab[i][e] += fg[e];
                    const float t = fg[e] * g[i][e];
                    gy[i][e] = t;
                    c1 += t;
                    c2 += t * h;
                }
            }
            c1 = group_sum<LPR>(c1) / p.C;
            c2 = group_sum<LPR>(c2) / p.C;
            if (rv) {
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lr + i * LPR;
                    if (v < nvec) {
                        float o[8], fa[8];
                        cvt_vec<TX>(R.ra[i], fa);
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            o[e] = R.rstd * (gy[i][e] - c1 - xh[i][e] * c2) + fa[e];
                            dsum[i][e] += o[e];
                            o[e] *= R.sc;
                        }
                        if (dbr) st_vec<TG, VE>(dbr + rbase + v * VE, o);
                    }
                }
            }
        };
        if constexpr (sizeof(TX) == 4) {  // two streams in flight (bf16 rows: 8 elements per vector, that needs > 256 VGPRs)
            Regs A, Bq;
            load(A, 0);
            for (int k = 0; k < p.nk; k += 2) {
                if (k + 1 < p.nk) load(Bq, k + 1);
                compute(A, k);
                if (k + 1 < p.nk) {
                    if (k + 2 < p.nk) load(A, k + 2);
                    compute(Bq, k + 1);
                }
            }
        } else {
            Regs A;
            for (int k = 0; k < p.nk; ++k) {
                load(A, k);
                compute(A, k);
            }
        }
        if (rv) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lr + i * LPR;
                if (v < nvec) st_vec<TX, VE>(dsh + rbase + v * VE, dsum[i]);
            }
        }
    }
    float* mine = sm + (size_t)(wave * RPW + sub) * 2 * p.C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
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

// second stage: one workgroup per 64 columns; its 16 waves stride over the partial rows (coalesced 256-byte reads, 4
// independent chains each: 8 dependent iterations for 512 partials instead of 32), then a fixed-order LDS combine ->
// deterministic
constexpr int LN_RW = 16;
__global__ __launch_bounds__(64 * LN_RW) void k_ln_reduce(const float* part, float* dgamma, float* dbeta, int nblk, int C) {
    __shared__ float sm[LN_RW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // column in [0, 2C)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < 2 * C) {
        int b = wave;
        for (; b + 3 * LN_RW < nblk; b += 4 * LN_RW) {
            a0 += part[(int64_t)b * 2 * C + c];
            a1 += part[(int64_t)(b + LN_RW) * 2 * C + c];
            a2 += part[(int64_t)(b + 2 * LN_RW) * 2 * C + c];
            a3 += part[(int64_t)(b + 3 * LN_RW) * 2 * C + c];
        }
        for (; b < nblk; b += LN_RW) a0 += part[(int64_t)b * 2 * C + c];
    }
    sm[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && c < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < LN_RW; ++w) t += sm[w][lane];
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

int ln_grid(int64_t M, int lpr, int cap = 256 * 4, int unr = 1) {
    const int64_t rows_per_blk = 4 * (64 / lpr) * unr;
    int64_t g = mtl_ceil_div(M, rows_per_blk);
    if (g > cap) g = cap;  // also the number of dgamma/dbeta partials the second stage sums
    return (int)(g < 1 ? 1 : g);
}

// patch-merging gather: rows = B * (H/2) * (W/2), C = 4 * C_token, C_token a multiple of the vector width
int ln_merge(LnParams& p, int64_t M, int64_t C, int xdt, int mh, int mw) {
    p.mg_H = p.mg_W = 0;
    if (mh == 0 && mw == 0) return MTLORA_OK;
    const int ve = xdt == MTLORA_F32 ? 4 : 8;
    if (mh <= 0 || mw <= 0 || (mh & 1) || (mw & 1) || C % 4 || (C / 4) % ve) return MTLORA_ERR_SHAPE;
    if (M % ((int64_t)(mh / 2) * (mw / 2))) return MTLORA_ERR_SHAPE;
    p.mg_H = mh;
    p.mg_W = mw;
    return MTLORA_OK;
}

int ln_check(int64_t M, int64_t C, int xdt, int ydt) {
    if (xdt < MTLORA_F32 || xdt > MTLORA_F16 || ydt < MTLORA_F32 || ydt > MTLORA_F16) return MTLORA_ERR_DTYPE;
    if (xdt != MTLORA_F32 && ydt != MTLORA_F32 && xdt != ydt) return MTLORA_ERR_DTYPE;  // (bf16 <-> fp16 mixes: none)
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

#define LN_LAUNCH(KERNEL, L_, ...)                                                                        \
    if (vpl <= 3)                                                                                           \
        hipLaunchKernelGGL((KERNEL<__VA_ARGS__, L_, 3 LN_EXTRA>), dim3(grid), dim3(256), lds, s, p);         \
    else                                                                                                    \
        hipLaunchKernelGGL((KERNEL<__VA_ARGS__, L_, LN_MAXV LN_EXTRA>), dim3(grid), dim3(256), lds, s, p);
#define LN_DISPATCH_LPR(KERNEL, ...)                       \
    switch (lpr) {                                         \
        case 8: LN_LAUNCH(KERNEL, 8, __VA_ARGS__) break;   \
        case 16: LN_LAUNCH(KERNEL, 16, __VA_ARGS__) break; \
        case 32: LN_LAUNCH(KERNEL, 32, __VA_ARGS__) break; \
        default: LN_LAUNCH(KERNEL, 64, __VA_ARGS__) break; \
    }

static int ln_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t M,
                       int64_t C, float eps, int x_dtype, int y_dtype, int merge_h, int merge_w, const void* branch,
                       void* x_new, const float* scale, int64_t B, void* stream) {
    int st = ln_check(M, C, x_dtype, y_dtype);
    if (st != MTLORA_OK) return st;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)branch | (uintptr_t)x_new) & 15u) return MTLORA_ERR_ALIGN;
    if (branch && (!x_new || B <= 0 || M % B || merge_h || merge_w)) return MTLORA_ERR_SHAPE;
    if (M == 0) return MTLORA_OK;
    LnParams p = {};
    p.rb = branch;
    p.xsum = x_new;
    p.rscale = scale;
    p.rows_per_sample = branch ? M / B : 1;
    p.x = x;
    p.gamma = gamma;
    p.beta = beta;
    p.y = y;
    p.mean = mean;
    p.rstd = rstd;
    p.M = M;
    p.C = (int)C;
    p.eps = eps;
    st = ln_merge(p, M, C, x_dtype, merge_h, merge_w);
    if (st != MTLORA_OK) return st;
    const int nvec_h = (int)(C / (x_dtype == MTLORA_F32 ? 4 : 8));
    const int lpr = pick_lpr(nvec_h);
    const int vpl = (nvec_h + lpr - 1) / lpr;
    const int grid = (int)mtl_ceil_div(ln_grid(M, lpr, 1 << 30), vpl <= 3 ? 4 : 1) < 256 * 8
                         ? (int)mtl_ceil_div(ln_grid(M, lpr, 1 << 30), vpl <= 3 ? 4 : 1)
                         : 256 * 8;  // 4 row groups per wave iteration in the narrow-row kernels
    const size_t lds = 0;
    hipStream_t s = (hipStream_t)stream;
    const int es_x = mtl_elem_size(x_dtype), es_y = mtl_elem_size(y_dtype);
    mtl_prof_tag("M%lld C%lld x%d y%d mg%d", (long long)M, (long long)C, x_dtype, y_dtype, merge_w);
    MtlProfScope prof(PK_LN_FWD, (double)M * C * (es_x + es_y + (branch ? es_x + es_y : 0)), s);
    if (branch) {
#define LN_EXTRA , true
        if (x_dtype == MTLORA_F32 && y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, float)
        } else if (x_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, float, f16)
            } else if (y_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, f16, float)
            } else {
                LN_DISPATCH_LPR(k_ln_fwd, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, bf16)
        } else if (y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, bf16)
        }
#undef LN_EXTRA
    } else {
#define LN_EXTRA , false
        if (x_dtype == MTLORA_F32 && y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, float)
        } else if (x_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, float, f16)
            } else if (y_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, f16, float)
            } else {
                LN_DISPATCH_LPR(k_ln_fwd, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, bf16)
        } else if (y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, bf16)
        }
#undef LN_EXTRA
    }
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t M, int64_t C, float eps, int x_dtype, int y_dtype, int merge_h, int merge_w, void* stream) {
    return ln_fwd_impl(x, gamma, beta, y, mean, rstd, M, C, eps, x_dtype, y_dtype, merge_h, merge_w, nullptr, nullptr, nullptr,
                       1, stream);
}

int mtlora_residual_layernorm_fwd(const void* shortcut, const void* branch, const float* scale, int64_t B, const float* gamma,
                                  const float* beta, void* x_new, void* y, float* mean, float* rstd, int64_t M, int64_t C,
                                  float eps, int x_dtype, int y_dtype, void* stream) {
    if (!branch || !x_new) return MTLORA_ERR_NULL;
    return ln_fwd_impl(shortcut, gamma, beta, y, mean, rstd, M, C, eps, x_dtype, y_dtype, 0, 0, branch, x_new, scale, B, stream);
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                       float* dgamma, float* dbeta, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                       int64_t scratch_bytes, const void* dx_addend, int merge_h, int merge_w, void* d_branch,
                       const float* scale, int64_t B, void* stream, int phase = 0) {
    // phase (internal.h): 0 main kernel + reduce, 1 main kernel only, 2 reduce of the partials in `scratch` only
    int st = ln_check(M, C, x_dtype, dy_dtype);
    if (st != MTLORA_OK) return st;
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !scratch) return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)scratch | (uintptr_t)dx_addend | (uintptr_t)d_branch) & 15u)
        return MTLORA_ERR_ALIGN;
    if (d_branch && (B <= 0 || M % B || merge_h || merge_w)) return MTLORA_ERR_SHAPE;
    if (scratch_bytes < mtlora_layernorm_bwd_scratch_bytes(M, C, x_dtype) - 256) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        mtl_zero_async(dgamma, (size_t)C * 4, s);
        mtl_zero_async(dbeta, (size_t)C * 4, s);
        return MTLORA_OK;
    }
    LnParams p = {};
    p.x = x;
    p.dy = dy;
    p.gamma = gamma;
    p.mean = const_cast<float*>(mean);
    p.rstd = const_cast<float*>(rstd);
    p.dx = dx;
    p.add = dx_addend;
    p.dbr = d_branch;
    p.rscale = scale;
    p.rows_per_sample = d_branch ? M / B : 1;
    p.part = reinterpret_cast<float*>(scratch);
    p.M = M;
    p.C = (int)C;
    st = ln_merge(p, M, C, x_dtype, merge_h, merge_w);
    if (st != MTLORA_OK) return st;
    const int nvec_h = (int)(C / (x_dtype == MTLORA_F32 ? 4 : 8));
    const int lpr = pick_lpr(nvec_h);
    const int vpl = (nvec_h + lpr - 1) / lpr;
    const int grid = ln_grid(M, lpr);
    const size_t lds = (size_t)4 * (64 / lpr) * 2 * C * 4;
    const int es_x = mtl_elem_size(x_dtype), es_g = mtl_elem_size(dy_dtype);
    if (phase != 2) {
        mtl_prof_tag("M%lld C%lld x%d g%d mg%d add%d", (long long)M, (long long)C, x_dtype, dy_dtype, merge_w, dx_addend ? 1 : 0);
        MtlProfScope prof(PK_LN_BWD, (double)M * C * (2 * es_x + es_g + (dx_addend ? es_x : 0) + (d_branch ? es_g : 0)), s);
#define LN_EXTRA
        if (x_dtype == MTLORA_F32 && dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, float, float)
        } else if (x_dtype == MTLORA_F16 || dy_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_bwd, float, f16)
            } else if (dy_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_bwd, f16, float)
            } else {
                LN_DISPATCH_LPR(k_ln_bwd, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, float, bf16)
        } else if (dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_bwd, bf16, bf16)
        }
#undef LN_EXTRA
    }
    if (phase != 1)
        hipLaunchKernelGGL(k_ln_reduce, dim3((unsigned)mtl_ceil_div(2 * C, 64)), dim3(64 * LN_RW), 0, s, (const float*)p.part, dgamma,
                           dbeta, grid, (int)C);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtli_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                       float* dgamma, float* dbeta, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                       int64_t scratch_bytes, const void* dx_addend, int phase, void* stream) {
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, M, C, x_dtype, dy_dtype, scratch, scratch_bytes, dx_addend, 0, 0,
                       nullptr, nullptr, 1, stream, phase);
}

int mtli_residual_layernorm_bwd(const void* dy, const void* x_new, const float* gamma, const float* mean, const float* rstd,
                                void* d_shortcut, void* d_branch, float* dgamma, float* dbeta, const float* scale, int64_t B,
                                int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch, int64_t scratch_bytes,
                                const void* dx_addend, int phase, void* stream) {
    if (!d_branch) return MTLORA_ERR_NULL;
    return ln_bwd_impl(dy, x_new, gamma, mean, rstd, d_shortcut, dgamma, dbeta, M, C, x_dtype, dy_dtype, scratch, scratch_bytes,
                       dx_addend, 0, 0, d_branch, scale, B, stream, phase);
}

int mtlora_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         void* dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int x_dtype, int dy_dtype,
                         void* scratch, int64_t scratch_bytes, const void* dx_addend, int merge_h, int merge_w, void* stream) {
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, M, C, x_dtype, dy_dtype, scratch, scratch_bytes, dx_addend,
                       merge_h, merge_w, nullptr, nullptr, 1, stream);
}

int mtlora_residual_layernorm_bwd(const void* dy, const void* x_new, const float* gamma, const float* mean, const float* rstd,
                                  void* d_shortcut, void* d_branch, float* dgamma, float* dbeta, const float* scale,
                                  int64_t B, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                                  int64_t scratch_bytes, const void* dx_addend, void* stream) {
    if (!d_branch) return MTLORA_ERR_NULL;
    return ln_bwd_impl(dy, x_new, gamma, mean, rstd, d_shortcut, dgamma, dbeta, M, C, x_dtype, dy_dtype, scratch, scratch_bytes,
                       dx_addend, 0, 0, d_branch, scale, B, stream);
}

/* n independent inputs through the SAME LayerNorm in one launch each way (PatchMerging's norm applied to the shared tensor and
 * to every task tensor, swin_transformer_mtlora.py:543-551): dgamma / dbeta come out summed over the inputs. */
int64_t mtlora_layernorm_multi_bwd_scratch_bytes(int n, int64_t M, int64_t C, int x_dtype) {
    const int64_t one = mtlora_layernorm_bwd_scratch_bytes(M, C, x_dtype);
    if (one < 0 || n < 1 || n > MTLORA_MAX_TASKS + 1) return -1;
    return (one - 256) * n + 256;
}

static int ln_multi_fwd_impl(int n, const void* const* x, const float* gamma, const float* beta, void* const* y,
                             float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype, int y_dtype,
                             int merge_h, int merge_w, const void* const* branch, void* const* x_new, const float* scale,
                             int64_t B, void* stream) {
    int st = ln_check(M, C, x_dtype, y_dtype);
    if (st != MTLORA_OK) return st;
    if (n < 1 || n > MTLORA_MAX_TASKS + 1) return MTLORA_ERR_SHAPE;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return MTLORA_ERR_NULL;
    if (branch && (!x_new || B <= 0 || M % B)) return MTLORA_ERR_SHAPE;
    LnParams p = {};
    for (int k = 0; k < n; ++k) {
        if (!x[k] || !y[k] || !mean[k] || !rstd[k]) return MTLORA_ERR_NULL;
        if (((uintptr_t)x[k] | (uintptr_t)y[k]) & 15u) return MTLORA_ERR_ALIGN;
        p.x_k[k] = x[k];
        p.y_k[k] = y[k];
        p.mean_k[k] = mean[k];
        p.rstd_k[k] = rstd[k];
        if (branch) {
            if (!branch[k] || !x_new[k]) return MTLORA_ERR_NULL;
            if (((uintptr_t)branch[k] | (uintptr_t)x_new[k]) & 15u) return MTLORA_ERR_ALIGN;
            p.rb_k[k] = branch[k];
            p.xsum_k[k] = x_new[k];
        }
    }
    if (M == 0) return MTLORA_OK;
    if (branch) {
        p.rb = branch[0];
        p.xsum = x_new[0];
        p.rscale = scale;
    }
    p.multi_x = 1;
    p.x = x[0];
    p.y = y[0];
    p.mean = mean[0];
    p.rstd = rstd[0];
    p.gamma = gamma;
    p.beta = beta;
    p.M = M;
    p.C = (int)C;
    p.eps = eps;
    p.rows_per_sample = branch ? M / B : 1;
    st = ln_merge(p, M, C, x_dtype, merge_h, merge_w);
    if (st != MTLORA_OK) return st;
    const int nvec_h = (int)(C / (x_dtype == MTLORA_F32 ? 4 : 8));
    const int lpr = pick_lpr(nvec_h);
    const int vpl = (nvec_h + lpr - 1) / lpr;
    const int gx = (int)mtl_ceil_div(ln_grid(M, lpr, 1 << 30), vpl <= 3 ? 4 : 1) < 256 * 8
                       ? (int)mtl_ceil_div(ln_grid(M, lpr, 1 << 30), vpl <= 3 ? 4 : 1)
                       : 256 * 8;
    const dim3 grid((unsigned)gx, (unsigned)n);
    const size_t lds = 0;
    hipStream_t s = (hipStream_t)stream;
    const int es_x = mtl_elem_size(x_dtype), es_y = mtl_elem_size(y_dtype);
    mtl_prof_tag("M%lld C%lld x%d y%d mg%d n%d res%d", (long long)M, (long long)C, x_dtype, y_dtype, merge_w, n, branch ? 1 : 0);
    MtlProfScope prof(PK_LN_FWD, (double)n * M * C * (es_x + es_y + (branch ? es_x + es_y : 0)), s);
    if (branch) {
#define LN_EXTRA , true
        if (x_dtype == MTLORA_F32 && y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, float)
        } else if (x_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, float, f16)
            } else if (y_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, f16, float)
            } else {
                LN_DISPATCH_LPR(k_ln_fwd, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, bf16)
        } else if (y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, bf16)
        }
#undef LN_EXTRA
    } else {
#define LN_EXTRA , false
        if (x_dtype == MTLORA_F32 && y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, float)
        } else if (x_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, float, f16)
            } else if (y_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_fwd, f16, float)
            } else {
                LN_DISPATCH_LPR(k_ln_fwd, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, bf16)
        } else if (y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_fwd, bf16, bf16)
        }
#undef LN_EXTRA
    }
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_layernorm_multi_fwd(int n, const void* const* x, const float* gamma, const float* beta, void* const* y,
                               float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype, int y_dtype,
                               int merge_h, int merge_w, void* stream) {
    return ln_multi_fwd_impl(n, x, gamma, beta, y, mean, rstd, M, C, eps, x_dtype, y_dtype, merge_h, merge_w, nullptr, nullptr,
                             nullptr, 1, stream);
}

/* n independent streams, each  x_new[k] = res[k] + scale[k][sample] * branch[k]  then the SAME LayerNorm (plain rows or the
 * PatchMerging gather): the MLP residual of the task-enabled block fused with the stage's PatchMerging norm. */
int mtlora_residual_layernorm_streams_fwd(int n, const void* const* res, const void* const* branch, const float* scale,
                                          int64_t B, const float* gamma, const float* beta, void* const* x_new, void* const* y,
                                          float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype,
                                          int y_dtype, int merge_h, int merge_w, void* stream) {
    if (!branch || !x_new) return MTLORA_ERR_NULL;
    return ln_multi_fwd_impl(n, res, gamma, beta, y, mean, rstd, M, C, eps, x_dtype, y_dtype, merge_h, merge_w, branch, x_new,
                             scale, B, stream);
}

static int ln_multi_bwd_impl(int n, const void* const* dy, const void* const* x, const float* gamma, const float* const* mean,
                             const float* const* rstd, void* const* dx, float* dgamma, float* dbeta, int64_t M, int64_t C,
                             int x_dtype, int dy_dtype, void* scratch, int64_t scratch_bytes, const void* const* dx_addend,
                             int merge_h, int merge_w, void* const* d_branch, const float* scale, int64_t B, void* stream) {
    int st = ln_check(M, C, x_dtype, dy_dtype);
    if (d_branch && (B <= 0 || M % B)) return MTLORA_ERR_SHAPE;
    if (st != MTLORA_OK) return st;
    if (n < 1 || n > MTLORA_MAX_TASKS + 1) return MTLORA_ERR_SHAPE;
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !scratch) return MTLORA_ERR_NULL;
    if ((uintptr_t)scratch & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_layernorm_multi_bwd_scratch_bytes(n, M, C, x_dtype) - 256) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        mtl_zero_async(dgamma, (size_t)C * 4, s);
        mtl_zero_async(dbeta, (size_t)C * 4, s);
        return MTLORA_OK;
    }
    LnParams p = {};
    for (int k = 0; k < n; ++k) {
        if (!dy[k] || !x[k] || !mean[k] || !rstd[k] || !dx[k]) return MTLORA_ERR_NULL;
        const void* ad = dx_addend ? dx_addend[k] : nullptr;
        if (((uintptr_t)dy[k] | (uintptr_t)x[k] | (uintptr_t)dx[k] | (uintptr_t)ad) & 15u) return MTLORA_ERR_ALIGN;
        p.dy_k[k] = dy[k];
        p.x_k[k] = x[k];
        p.dx_k[k] = dx[k];
        p.add_k[k] = ad;
        p.mean_k[k] = const_cast<float*>(mean[k]);
        p.rstd_k[k] = const_cast<float*>(rstd[k]);
        if (d_branch) {
            if ((uintptr_t)d_branch[k] & 15u) return MTLORA_ERR_ALIGN;
            p.dbr_k[k] = d_branch[k];
        }
    }
    p.multi_x = 1;
    p.x = x[0];
    p.dy = dy[0];
    p.dx = dx[0];
    p.gamma = gamma;
    p.mean = const_cast<float*>(mean[0]);
    p.rstd = const_cast<float*>(rstd[0]);
    p.part = reinterpret_cast<float*>(scratch);
    p.M = M;
    p.C = (int)C;
    p.rscale = d_branch ? scale : nullptr;
    p.rows_per_sample = d_branch ? M / B : 1;
    st = ln_merge(p, M, C, x_dtype, merge_h, merge_w);
    if (st != MTLORA_OK) return st;
    const int nvec_h = (int)(C / (x_dtype == MTLORA_F32 ? 4 : 8));
    const int lpr = pick_lpr(nvec_h);
    const int vpl = (nvec_h + lpr - 1) / lpr;
    const int gx = ln_grid(M, lpr);
    const dim3 grid((unsigned)gx, (unsigned)n);
    const size_t lds = (size_t)4 * (64 / lpr) * 2 * C * 4;
    const int es_x = mtl_elem_size(x_dtype), es_g = mtl_elem_size(dy_dtype);
    {
        mtl_prof_tag("M%lld C%lld x%d g%d mg%d n%d", (long long)M, (long long)C, x_dtype, dy_dtype, merge_w, n);
        MtlProfScope prof(PK_LN_BWD, (double)n * M * C * (2 * es_x + es_g + (dx_addend ? es_x : 0)), s);
#define LN_EXTRA
        if (x_dtype == MTLORA_F32 && dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, float, float)
        } else if (x_dtype == MTLORA_F16 || dy_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_bwd, float, f16)
            } else if (dy_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_ln_bwd, f16, float)
            } else {
                LN_DISPATCH_LPR(k_ln_bwd, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, float, bf16)
        } else if (dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_bwd, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_bwd, bf16, bf16)
        }
#undef LN_EXTRA
    }
    hipLaunchKernelGGL(k_ln_reduce, dim3((unsigned)mtl_ceil_div(2 * C, 64)), dim3(64 * LN_RW), 0, s, (const float*)p.part, dgamma,
                       dbeta, gx * n, (int)C);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_layernorm_multi_bwd(int n, const void* const* dy, const void* const* x, const float* gamma, const float* const* mean,
                               const float* const* rstd, void* const* dx, float* dgamma, float* dbeta, int64_t M, int64_t C,
                               int x_dtype, int dy_dtype, void* scratch, int64_t scratch_bytes, const void* const* dx_addend,
                               int merge_h, int merge_w, void* stream) {
    return ln_multi_bwd_impl(n, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, M, C, x_dtype, dy_dtype, scratch, scratch_bytes,
                             dx_addend, merge_h, merge_w, nullptr, nullptr, 1, stream);
}

/* backward of mtlora_residual_layernorm_streams_fwd: d_res[k] = dx_addend[k] + LN-backward(dy[k]) (layout of res),
 * d_branch[k] = scale[k][sample] * d_res[k]; dgamma / dbeta summed over the streams. */
int mtlora_residual_layernorm_streams_bwd(int n, const void* const* dy, const void* const* x_new, const float* gamma,
                                          const float* const* mean, const float* const* rstd, void* const* d_res,
                                          void* const* d_branch, float* dgamma, float* dbeta, const float* scale, int64_t B,
                                          int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch, int64_t scratch_bytes,
                                          const void* const* dx_addend, int merge_h, int merge_w, void* stream) {
    if (!d_branch) return MTLORA_ERR_NULL;
    return ln_multi_bwd_impl(n, dy, x_new, gamma, mean, rstd, d_res, dgamma, dbeta, M, C, x_dtype, dy_dtype, scratch,
                             scratch_bytes, dx_addend, merge_h, merge_w, d_branch, scale, B, stream);
}

/* multi-stream forms: ONE shortcut, n branches -> n (x_new, y) pairs (task-enabled Swin block: swin_transformer_mtlora.py:389-396
 * for the shared stream and every task stream) */
int mtlora_residual_layernorm_multi_fwd(int n, const void* shortcut, const void* const* branch, const float* scale, int64_t B,
                                        const float* gamma, const float* beta, void* const* x_new, void* const* y,
                                        float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype,
                                        int y_dtype, void* stream) {
    int st = ln_check(M, C, x_dtype, y_dtype);
    if (st != MTLORA_OK) return st;
    if (n < 1 || n > MTLORA_MAX_TASKS + 1 || B <= 0 || M % B) return MTLORA_ERR_SHAPE;
    if (!shortcut || !branch || !gamma || !beta || !x_new || !y || !mean || !rstd) return MTLORA_ERR_NULL;
    if ((uintptr_t)shortcut & 15u) return MTLORA_ERR_ALIGN;
    LnParams p = {};
    for (int k = 0; k < n; ++k) {
        if (!branch[k] || !x_new[k] || !y[k] || !mean[k] || !rstd[k]) return MTLORA_ERR_NULL;
        if (((uintptr_t)branch[k] | (uintptr_t)x_new[k] | (uintptr_t)y[k]) & 15u) return MTLORA_ERR_ALIGN;
        p.rb_k[k] = branch[k];
        p.xsum_k[k] = x_new[k];
        p.y_k[k] = y[k];
        p.mean_k[k] = mean[k];
        p.rstd_k[k] = rstd[k];
    }
    if (M == 0) return MTLORA_OK;
    p.nk = n;
    p.x = shortcut;
    p.gamma = gamma;
    p.beta = beta;
    p.rb = branch[0];
    p.xsum = x_new[0];
    p.y = y[0];
    p.mean = mean[0];
    p.rstd = rstd[0];
    p.rscale = scale;
    p.rows_per_sample = M / B;
    p.M = M;
    p.C = (int)C;
    p.eps = eps;
    const int nvec_h = (int)(C / (x_dtype == MTLORA_F32 ? 4 : 8));
    const int lpr = pick_lpr(nvec_h);
    const int vpl = (nvec_h + lpr - 1) / lpr;
    const int gx = (int)mtl_ceil_div(ln_grid(M, lpr, 1 << 30), vpl <= 3 ? 4 : 1) < 256 * 8
                       ? (int)mtl_ceil_div(ln_grid(M, lpr, 1 << 30), vpl <= 3 ? 4 : 1)
                       : 256 * 8;
    const dim3 grid((unsigned)gx, (unsigned)n);
    const size_t lds = 0;
    hipStream_t s = (hipStream_t)stream;
    const int es_x = mtl_elem_size(x_dtype), es_y = mtl_elem_size(y_dtype);
    mtl_prof_tag("M%lld C%lld x%d y%d n%d", (long long)M, (long long)C, x_dtype, y_dtype, n);
    MtlProfScope prof(PK_LN_FWD, (double)M * C * (es_x + (double)n * (es_x + 2 * es_y)), s);
#define LN_EXTRA , true
    if (x_dtype == MTLORA_F32 && y_dtype == MTLORA_F32) {
        LN_DISPATCH_LPR(k_ln_fwd, float, float)
    } else if (x_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
        if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, float, f16)
        } else if (y_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_ln_fwd, f16, float)
        } else {
            LN_DISPATCH_LPR(k_ln_fwd, f16, f16)
        }
    } else if (x_dtype == MTLORA_F32) {
        LN_DISPATCH_LPR(k_ln_fwd, float, bf16)
    } else if (y_dtype == MTLORA_F32) {
        LN_DISPATCH_LPR(k_ln_fwd, bf16, float)
    } else {
        LN_DISPATCH_LPR(k_ln_fwd, bf16, bf16)
    }
#undef LN_EXTRA
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

/* d_shortcut = sum_k (dx_addend[k] + LN-backward(dy[k]));  d_branch[k] = scale[k][sample] * (dx_addend[k] + LN-backward(dy[k]));
 * dgamma / dbeta summed over the streams.  dx_addend[k] / d_branch[k] may be NULL. */
int mtlora_residual_layernorm_multi_bwd(int n, const void* const* dy, const void* const* x_new, const float* gamma,
                                        const float* const* mean, const float* const* rstd, const void* const* dx_addend,
                                        void* d_shortcut, void* const* d_branch, float* dgamma, float* dbeta, const float* scale,
                                        int64_t B, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                                        int64_t scratch_bytes, void* stream) {
    int st = ln_check(M, C, x_dtype, dy_dtype);
    if (st != MTLORA_OK) return st;
    if (n < 1 || n > MTLORA_MAX_TASKS + 1 || B <= 0 || M % B) return MTLORA_ERR_SHAPE;
    if (!dy || !x_new || !gamma || !mean || !rstd || !d_shortcut || !dgamma || !dbeta || !scratch) return MTLORA_ERR_NULL;
    if (((uintptr_t)d_shortcut | (uintptr_t)scratch) & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_layernorm_bwd_scratch_bytes(M, C, x_dtype) - 256) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        mtl_zero_async(dgamma, (size_t)C * 4, s);
        mtl_zero_async(dbeta, (size_t)C * 4, s);
        return MTLORA_OK;
    }
    LnParams p = {};
    for (int k = 0; k < n; ++k) {
        if (!dy[k] || !x_new[k] || !mean[k] || !rstd[k]) return MTLORA_ERR_NULL;
        const void* ad = dx_addend ? dx_addend[k] : nullptr;
        void* db = d_branch ? d_branch[k] : nullptr;
        if (((uintptr_t)dy[k] | (uintptr_t)x_new[k] | (uintptr_t)ad | (uintptr_t)db) & 15u) return MTLORA_ERR_ALIGN;
        p.dy_k[k] = dy[k];
        p.xsum_k[k] = const_cast<void*>(x_new[k]);
        p.mean_k[k] = const_cast<float*>(mean[k]);
        p.rstd_k[k] = const_cast<float*>(rstd[k]);
        p.add_k[k] = ad;
        p.dbr_k[k] = db;
    }
    p.nk = n;
    p.gamma = gamma;
    p.dx = d_shortcut;
    p.rscale = scale;
    p.rows_per_sample = M / B;
    p.part = reinterpret_cast<float*>(scratch);
    p.M = M;
    p.C = (int)C;
    const int nvec_h = (int)(C / (x_dtype == MTLORA_F32 ? 4 : 8));
    const int lpr = pick_lpr(nvec_h);
    const int vpl = (nvec_h + lpr - 1) / lpr;
    const int grid = ln_grid(M, lpr);
    const size_t lds = (size_t)4 * (64 / lpr) * 2 * C * 4;
    const int es_x = mtl_elem_size(x_dtype), es_g = mtl_elem_size(dy_dtype);
    {
        mtl_prof_tag("M%lld C%lld x%d g%d n%d", (long long)M, (long long)C, x_dtype, dy_dtype, n);
        MtlProfScope prof(PK_LN_BWD, (double)M * C * (es_x + (double)n * (2 * es_x + 2 * es_g)), s);
#define LN_EXTRA
        if (x_dtype == MTLORA_F32 && dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_resln_bwd_multi, float, float)
        } else if (x_dtype == MTLORA_F16 || dy_dtype == MTLORA_F16) {  // fp16 autocast (the reference's default, main.py:341)
            if (x_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_resln_bwd_multi, float, f16)
            } else if (dy_dtype == MTLORA_F32) {
                LN_DISPATCH_LPR(k_resln_bwd_multi, f16, float)
            } else {
                LN_DISPATCH_LPR(k_resln_bwd_multi, f16, f16)
            }
        } else if (x_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_resln_bwd_multi, float, bf16)
        } else if (dy_dtype == MTLORA_F32) {
            LN_DISPATCH_LPR(k_resln_bwd_multi, bf16, float)
        } else {
            LN_DISPATCH_LPR(k_resln_bwd_multi, bf16, bf16)
        }
#undef LN_EXTRA
    }
    hipLaunchKernelGGL(k_ln_reduce, dim3((unsigned)mtl_ceil_div(2 * C, 64)), dim3(64 * LN_RW), 0, s, (const float*)p.part, dgamma,
                       dbeta, grid, (int)C);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}
}

// =================================================================================================
// BatchNorm (training) + optional ReLU over a channels-last (R rows x C channels) matrix -- the decoder heads'
// conv1x1 -> BN -> ReLU (seg_hrnet.py:498-526) evaluated on the (pixels, 1080) matrix.  Replaces ATen's
// batch_norm_*_channels_last kernels + the separate ReLU pass (13 ms/step at ~0.6 TB/s on MI355X).
//   forward : k_bn_stats (per-workgroup column sums / sums of squares -> (count, mean, M2) partials)
//             k_bn_finalize (Chan combine -> mean, rstd, scale = gamma*rstd, shift = beta - mean*scale, running stats)
//             k_bn_apply    (y = relu(x*scale + shift), 16 B per lane)
//   backward: k_bn_bwd_stats (sum dy', sum dy'*xhat with dy' = dy * [y > 0]) -> k_bn_bwd_finalize (dgamma, dbeta)
//             k_bn_bwd_apply (dx = scale * (dy' - mean(dy') - xhat * mean(dy' xhat)))
// A thread owns ONE 16-byte channel vector for all rows it visits (fixed column -> register accumulators).
// =================================================================================================
namespace {

struct BnParams {
    const void* x;
    const void* dy;
    void* y;
    void* dx;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* mean;    // (C)
    float* rstd;    // (C)
    float* scale;   // (C) gamma * rstd
    float* shift;   // (C) beta - mean * scale
    float* part;    // [nblk][2][C] (+ counts)
    float* dgamma;
    float* dbeta;
    float* c1;      // (C) mean(dy')
    float* c2;      // (C) mean(dy' xhat)
    int64_t R;
    int C, VW, rows_per_iter, nblk, relu;
    float momentum, eps;
};

// column sums over the rows of this workgroup: s0 = sum f0(row), s1 = sum f1(row)
template <typename T, bool BWD>
__global__ __launch_bounds__(512) void k_bn_colsum(const BnParams p) {
    constexpr int VE = ET<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [rows_per_iter][2][C]
    const int tid = threadIdx.x;
    const int ro = tid / p.VW, v = tid % p.VW;
    const bool active = ro < p.rows_per_iter;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* dy = reinterpret_cast<const T*>(p.dy);
    float a0[VE], a1[VE], sc[VE], sh[VE], mu[VE], rs[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        a0[e] = a1[e] = 0.f;
        if (BWD && active) {
            sc[e] = p.scale[v * VE + e];
            sh[e] = p.shift[v * VE + e];
            mu[e] = p.mean[v * VE + e];
            rs[e] = p.rstd[v * VE + e];
        }
    }
    const int64_t rows_blk = mtl_ceil_div(p.R, p.nblk);
    const int64_t r_lo = (int64_t)blockIdx.x * rows_blk;
    int64_t r_hi = r_lo + rows_blk;
    if (r_hi > p.R) r_hi = p.R;
    if (active) {
        constexpr int UN = 4;  // rows in flight per thread (one 16-byte load each; a single load per iteration ran at 2 TB/s)
        for (int64_t r = r_lo + ro; r < r_hi; r += (int64_t)UN * p.rows_per_iter) {
            u32x4 rx[UN], rg[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t rr = r + (int64_t)u * p.rows_per_iter;
                const bool ok = rr < r_hi;
                rx[u] = ok ? *reinterpret_cast<const u32x4*>(x + rr * p.C + v * VE) : u32x4{0u, 0u, 0u, 0u};
                if (BWD) rg[u] = ok ? *reinterpret_cast<const u32x4*>(dy + rr * p.C + v * VE) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float fx[8];
                cvt_vec<T>(rx[u], fx);
                if (!BWD) {  // rows past the end were loaded as zeros: they add nothing
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        a0[e] += fx[e];
                        a1[e] += fx[e] * fx[e];
                    }
                } else {
                    float fg[8];
                    cvt_vec<T>(rg[u], fg);  // (zero gradient for rows past the end)
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        const float yv = fx[e] * sc[e] + sh[e];
                        const float g = (p.relu && yv <= 0.f) ? 0.f : fg[e];
                        a0[e] += g;
                        a1[e] += g * (fx[e] - mu[e]) * rs[e];
                    }
                }
            }
        }
        float* mine = sm + (size_t)ro * 2 * p.C;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            mine[v * VE + e] = a0[e];
            mine[p.C + v * VE + e] = a1[e];
        }
    }
    __syncthreads();
    float* dst = p.part + (int64_t)blockIdx.x * 2 * p.C;
    for (int i = tid; i < 2 * p.C; i += blockDim.x) {
        float t = 0.f;
        for (int g = 0; g < p.rows_per_iter; ++g) t += sm[(size_t)g * 2 * p.C + i];
        dst[i] = t;
    }
}

// forward finalize: per channel combine (count, sum, sumsq) partials with Chan's formula.
// One workgroup per 64 channels: its 4 waves each fold a strided quarter of the partials (coalesced 256-byte reads),
// then wave 0 folds the four results in a fixed order.
// Finalize kernels: the per-workgroup partials (fp32 sums of x and x^2, resp. of dy' and dy' xhat) are summed in fp64 by
// 16 waves with two independent chains each and combined in a fixed order (deterministic).  Plain fp64 sums carry the
// same information as a Chan fold of the fp32 partials (var = (Q - S^2/R)/R loses < 1e-12 relative in fp64) without the
// two fp64 divisions per partial that made the fold take 29-67 us.
constexpr int BN_FW = 16;
__global__ __launch_bounds__(64 * BN_FW) void k_bn_finalize(const BnParams p) {
    __shared__ double sm[BN_FW][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
    if (c < p.C) {
        int b = wave;
        for (; b + BN_FW < p.nblk; b += 2 * BN_FW) {
            s0 += p.part[(int64_t)b * 2 * p.C + c];
            q0 += p.part[(int64_t)b * 2 * p.C + p.C + c];
            s1 += p.part[(int64_t)(b + BN_FW) * 2 * p.C + c];
            q1 += p.part[(int64_t)(b + BN_FW) * 2 * p.C + p.C + c];
        }
        for (; b < p.nblk; b += BN_FW) {
            s0 += p.part[(int64_t)b * 2 * p.C + c];
            q0 += p.part[(int64_t)b * 2 * p.C + p.C + c];
        }
    }
    sm[wave][0][lane] = s0 + s1;
    sm[wave][1][lane] = q0 + q1;
    __syncthreads();
    if (wave != 0 || c >= p.C) return;
    double S = 0.0, Q = 0.0;
    for (int w = 0; w < BN_FW; ++w) {
        S += sm[w][0][lane];
        Q += sm[w][1][lane];
    }
    const double n = (double)p.R, mean = S / n;
    double m2 = Q - S * mean;
    m2 = m2 < 0.0 ? 0.0 : m2;
    const double var = m2 / n;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float sc = p.gamma[c] * rstd;
    p.mean[c] = (float)mean;
    p.rstd[c] = rstd;
    p.scale[c] = sc;
    p.shift[c] = p.beta[c] - (float)mean * sc;
    if (p.running_mean) {
        const double unb = n > 1.0 ? m2 / (n - 1.0) : var;
        p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
        p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unb;
    }
}

__global__ __launch_bounds__(64 * BN_FW) void k_bn_bwd_finalize(const BnParams p) {
    __shared__ double sm[BN_FW][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
    if (c < p.C) {
        int b = wave;
        for (; b + BN_FW < p.nblk; b += 2 * BN_FW) {
            s0 += p.part[(int64_t)b * 2 * p.C + c];
            s1 += p.part[(int64_t)b * 2 * p.C + p.C + c];
            t0 += p.part[(int64_t)(b + BN_FW) * 2 * p.C + c];
            t1 += p.part[(int64_t)(b + BN_FW) * 2 * p.C + p.C + c];
        }
        for (; b < p.nblk; b += BN_FW) {
            s0 += p.part[(int64_t)b * 2 * p.C + c];
            s1 += p.part[(int64_t)b * 2 * p.C + p.C + c];
        }
    }
    sm[wave][0][lane] = s0 + t0;
    sm[wave][1][lane] = s1 + t1;
    __syncthreads();
    if (wave != 0 || c >= p.C) return;
    s0 = s1 = 0.0;
    for (int w = 0; w < BN_FW; ++w) {
        s0 += sm[w][0][lane];
        s1 += sm[w][1][lane];
    }
    p.dbeta[c] = (float)s0;
    p.dgamma[c] = (float)s1;
    p.c1[c] = (float)(s0 / (double)p.R);
    p.c2[c] = (float)(s1 / (double)p.R);
}

// a thread owns ONE channel vector (its per-channel constants stay in registers -- re-reading 2-6 per-channel arrays per
// element through the vector cache held the backward at 3.4 TB/s) and walks the rows of its workgroup's slab, 4 in flight
constexpr int BN_APPLY_BLOCKS = 2048;
template <typename T, bool BWD>
__global__ __launch_bounds__(512) void k_bn_apply(const BnParams p) {
    constexpr int VE = ET<T>::VEC;
    constexpr int UN = 4;
    const int tid = threadIdx.x;
    const int ro = tid / p.VW, v = tid % p.VW;
    if (ro >= p.rows_per_iter) return;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* dy = reinterpret_cast<const T*>(p.dy);
    T* out = reinterpret_cast<T*>(BWD ? p.dx : p.y);
    // forward: y = relu(x * sc + sh);  backward: dx = sc * g + kx * x + k0 with g = dy * [x * sc + sh > 0]
    float sc[VE], sh[VE], kx[VE], k0[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int c = v * VE + e;
        sc[e] = p.scale[c];
        sh[e] = p.shift[c];
        if (BWD) {
            const float t = sc[e] * p.c2[c] * p.rstd[c];
            kx[e] = -t;
            k0[e] = t * p.mean[c] - sc[e] * p.c1[c];
        }
    }
    const int64_t rows_blk = mtl_ceil_div(p.R, (int64_t)gridDim.x);
    const int64_t r_lo = (int64_t)blockIdx.x * rows_blk;
    int64_t r_hi = r_lo + rows_blk;
    if (r_hi > p.R) r_hi = p.R;
    for (int64_t r = r_lo + ro; r < r_hi; r += (int64_t)UN * p.rows_per_iter) {
        u32x4 rx[UN], rg[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t rr = r + (int64_t)u * p.rows_per_iter;
            const bool ok = rr < r_hi;
            rx[u] = ok ? *reinterpret_cast<const u32x4*>(x + rr * p.C + v * VE) : u32x4{0u, 0u, 0u, 0u};
            if (BWD) rg[u] = ok ? *reinterpret_cast<const u32x4*>(dy + rr * p.C + v * VE) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t rr = r + (int64_t)u * p.rows_per_iter;
            if (rr >= r_hi) continue;
            float fx[8], o[8];
            cvt_vec<T>(rx[u], fx);
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float yv = fx[e] * sc[e] + sh[e];
                    o[e] = (p.relu && yv < 0.f) ? 0.f : yv;
                }
            } else {
                float fg[8];
                cvt_vec<T>(rg[u], fg);
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float yv = fx[e] * sc[e] + sh[e];
                    const float g = (p.relu && yv <= 0.f) ? 0.f : fg[e];
                    o[e] = sc[e] * g + kx[e] * fx[e] + k0[e];
                }
            }
            st_vec<T, VE>(out + rr * p.C + v * VE, o);
        }
    }
}

int bn_setup(BnParams& p, int64_t R, int64_t C, int dtype) {
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16 && dtype != MTLORA_F16) return MTLORA_ERR_DTYPE;
    const int ve = dtype == MTLORA_F32 ? 4 : 8;
    if (R <= 0 || C <= 0 || C % ve) return MTLORA_ERR_SHAPE;
    p.R = R;
    p.C = (int)C;
    p.VW = (int)(C / ve);
    if (p.VW > 512) return MTLORA_ERR_UNSUPPORTED;
    p.rows_per_iter = 512 / p.VW;
    if (p.rows_per_iter > 8) p.rows_per_iter = 8;
    int64_t nblk = mtl_ceil_div(R, 64);
    if (nblk > 512) nblk = 512;
    p.nblk = (int)nblk;
    return MTLORA_OK;
}

}  // namespace

extern "C" {

// scratch: [nblk][2][C] partials + 4 C floats (scale, shift, c1, c2)
int64_t mtlora_bn_scratch_bytes(int64_t R, int64_t C, int dtype) {
    BnParams p = {};
    if (bn_setup(p, R, C, dtype) != MTLORA_OK) return -1;
    return ((int64_t)p.nblk * 2 * C + 4 * C) * 4 + 256;
}

int mtlora_bn_relu_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, int relu, void* y, float* save_mean, float* save_rstd,
                       float* save_scale, float* save_shift, int64_t R, int64_t C, int dtype, void* scratch,
                       int64_t scratch_bytes, void* stream) {
    BnParams p = {};
    int st = bn_setup(p, R, C, dtype);
    if (st != MTLORA_OK) return st;
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !save_scale || !save_shift || !scratch)
        return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)scratch) & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_bn_scratch_bytes(R, C, dtype) - 256) return MTLORA_ERR_WORKSPACE;
    p.x = x;
    p.y = y;
    p.gamma = gamma;
    p.beta = beta;
    p.running_mean = running_mean;
    p.running_var = running_var;
    p.mean = save_mean;
    p.rstd = save_rstd;
    p.scale = save_scale;
    p.shift = save_shift;
    p.part = reinterpret_cast<float*>(scratch);
    p.momentum = momentum;
    p.eps = eps;
    p.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    const int threads = (int)mtl_round_up((int64_t)p.rows_per_iter * p.VW, 64);
    const unsigned apply_blocks = (unsigned)(mtl_ceil_div(R, 16) < BN_APPLY_BLOCKS ? mtl_ceil_div(R, 16) : BN_APPLY_BLOCKS);
    const size_t lds = (size_t)p.rows_per_iter * 2 * C * 4;
    const int es = mtl_elem_size(dtype);
    {
        MtlProfScope prof(PK_BN, (double)R * C * es, s);
        if (dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_bn_colsum<float, false>), dim3(p.nblk), dim3(threads), lds, s, p);
        else if (dtype == MTLORA_F16)
            hipLaunchKernelGGL((k_bn_colsum<f16, false>), dim3(p.nblk), dim3(threads), lds, s, p);
        else
            hipLaunchKernelGGL((k_bn_colsum<bf16, false>), dim3(p.nblk), dim3(threads), lds, s, p);
    }
    hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)mtl_ceil_div(C, 64)), dim3(64 * BN_FW), 0, s, p);
    {
        MtlProfScope prof(PK_BN, (double)R * C * es * 2, s);
        if (dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_bn_apply<float, false>), dim3(apply_blocks), dim3(threads), 0, s, p);
        else if (dtype == MTLORA_F16)
            hipLaunchKernelGGL((k_bn_apply<f16, false>), dim3(apply_blocks), dim3(threads), 0, s, p);
        else
            hipLaunchKernelGGL((k_bn_apply<bf16, false>), dim3(apply_blocks), dim3(threads), 0, s, p);
    }
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_bn_relu_bwd(const void* dy, const void* x, const float* save_mean, const float* save_rstd,
                       const float* save_scale, const float* save_shift, int relu, void* dx, float* dgamma, float* dbeta,
                       int64_t R, int64_t C, int dtype, void* scratch, int64_t scratch_bytes, void* stream) {
    BnParams p = {};
    int st = bn_setup(p, R, C, dtype);
    if (st != MTLORA_OK) return st;
    if (!dy || !x || !save_mean || !save_rstd || !save_scale || !save_shift || !dx || !dgamma || !dbeta || !scratch)
        return MTLORA_ERR_NULL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)scratch) & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_bn_scratch_bytes(R, C, dtype) - 256) return MTLORA_ERR_WORKSPACE;
    p.x = x;
    p.dy = dy;
    p.dx = dx;
    p.mean = const_cast<float*>(save_mean);
    p.rstd = const_cast<float*>(save_rstd);
    p.scale = const_cast<float*>(save_scale);
    p.shift = const_cast<float*>(save_shift);
    p.part = reinterpret_cast<float*>(scratch);
    p.c1 = p.part + (int64_t)p.nblk * 2 * C;
    p.c2 = p.c1 + C;
    p.dgamma = dgamma;
    p.dbeta = dbeta;
    p.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    const int threads = (int)mtl_round_up((int64_t)p.rows_per_iter * p.VW, 64);
    const unsigned apply_blocks = (unsigned)(mtl_ceil_div(R, 16) < BN_APPLY_BLOCKS ? mtl_ceil_div(R, 16) : BN_APPLY_BLOCKS);
    const size_t lds = (size_t)p.rows_per_iter * 2 * C * 4;
    const int es = mtl_elem_size(dtype);
    {
        MtlProfScope prof(PK_BN, (double)R * C * es * 2, s);
        if (dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_bn_colsum<float, true>), dim3(p.nblk), dim3(threads), lds, s, p);
        else if (dtype == MTLORA_F16)
            hipLaunchKernelGGL((k_bn_colsum<f16, true>), dim3(p.nblk), dim3(threads), lds, s, p);
        else
            hipLaunchKernelGGL((k_bn_colsum<bf16, true>), dim3(p.nblk), dim3(threads), lds, s, p);
    }
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((unsigned)mtl_ceil_div(C, 64)), dim3(64 * BN_FW), 0, s, p);
    {
        MtlProfScope prof(PK_BN, (double)R * C * es * 3, s);
        if (dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_bn_apply<float, true>), dim3(apply_blocks), dim3(threads), 0, s, p);
        else if (dtype == MTLORA_F16)
            hipLaunchKernelGGL((k_bn_apply<f16, true>), dim3(apply_blocks), dim3(threads), 0, s, p);
        else
            hipLaunchKernelGGL((k_bn_apply<bf16, true>), dim3(apply_blocks), dim3(threads), 0, s, p);
    }
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}
}

// =================================================================================================
// residual + DropPath over the 1+T tensors of a block half (swin_transformer_mtlora.py:389-392, 398-408):
//   out_k[m] = res_k[m] + s_k[sample(m)] * y_k[m]        s = DropPath mask / keep (1 when off)
// forward : ONE launch for all k (reads res_k, y_k once, writes out_k) instead of a mul + an add per tensor;
// backward: ONE launch: dy_k = s_k * g_k and, when the residual is SHARED by all k (attention half: every task
//           output adds the same shortcut), d_res = sum_k g_k -- instead of T adds by the autograd engine.
// res / out dtype may be fp32 while y is bf16 (the stage-0 residual stream is fp32 under autocast).
// =================================================================================================
namespace {

struct ResParams {
    const void* res[MTLORA_MAX_TASKS + 1];
    const void* y[MTLORA_MAX_TASKS + 1];   // forward: y_k ; backward: g_k (dtype of out)
    void* out[MTLORA_MAX_TASKS + 1];       // forward: out_k ; backward: dy_k
    void* dres;                            // backward, shared residual: sum_k g_k
    const float* scale;                    // [n][B] or null (all ones)
    int64_t M;
    int C, n, B;
    int64_t rows_per_sample;
};

// TR: dtype of res / out / g ; TY: dtype of y / dy
template <typename TR, typename TY>
__global__ __launch_bounds__(256) void k_residual_fwd(const ResParams p) {
    const int64_t nvec = p.M * p.C / 8;
    const int k = blockIdx.y;
    const TR* res = reinterpret_cast<const TR*>(p.res[k]);
    const TY* y = reinterpret_cast<const TY*>(p.y[k]);
    TR* out = reinterpret_cast<TR*>(p.out[k]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = (i * 8) / p.C;
        const float s = p.scale ? p.scale[(int64_t)k * p.B + row / p.rows_per_sample] : 1.f;
        float fr[8], fy[8], o[8];
        if constexpr (sizeof(TR) == 4) {
            ld_vec<TR>(res + i * 8, fr);
            float hi[8];
            ld_vec<TR>(res + i * 8 + 4, hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) fr[4 + e] = hi[e];
        } else {
            ld_vec<TR>(res + i * 8, fr);
        }
        if constexpr (sizeof(TY) == 4) {
            ld_vec<TY>(y + i * 8, fy);
            float hi[8];
            ld_vec<TY>(y + i * 8 + 4, hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) fy[4 + e] = hi[e];
        } else {
            ld_vec<TY>(y + i * 8, fy);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fr[e] + s * fy[e];
        st_vec<TR, 8>(out + i * 8, o);
    }
}

template <typename TR, typename TY>
__global__ __launch_bounds__(256) void k_residual_bwd(const ResParams p) {
    const int64_t nvec = p.M * p.C / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = (i * 8) / p.C;
        const int64_t b = row / p.rows_per_sample;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int k = 0; k < p.n; ++k) {
            const TR* g = reinterpret_cast<const TR*>(p.y[k]);
            if (!g) continue;
            float fg[8];
            if constexpr (sizeof(TR) == 4) {
                ld_vec<TR>(g + i * 8, fg);
                float hi[8];
                ld_vec<TR>(g + i * 8 + 4, hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) fg[4 + e] = hi[e];
            } else {
                ld_vec<TR>(g + i * 8, fg);
            }
            const float s = p.scale ? p.scale[(int64_t)k * p.B + b] : 1.f;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] += fg[e];
                o[e] = s * fg[e];
            }
            if (p.out[k]) st_vec<TY, 8>(reinterpret_cast<TY*>(p.out[k]) + i * 8, o);
        }
        if (p.dres) st_vec<TR, 8>(reinterpret_cast<TR*>(p.dres) + i * 8, acc);
    }
}

int res_check(int n, int64_t M, int64_t C, int64_t B, int rdt, int ydt) {
    if (rdt < MTLORA_F32 || rdt > MTLORA_F16 || ydt < MTLORA_F32 || ydt > MTLORA_F16) return MTLORA_ERR_DTYPE;
    if (rdt != MTLORA_F32 && ydt != MTLORA_F32 && rdt != ydt) return MTLORA_ERR_DTYPE;
    if (n < 1 || n > MTLORA_MAX_TASKS + 1 || M < 0 || C <= 0 || C % 8 || B <= 0 || M % B) return MTLORA_ERR_SHAPE;
    return MTLORA_OK;
}

}  // namespace

extern "C" {

int mtlora_residual_droppath_fwd(int n, const void* const* res, const void* const* y, void* const* out,
                                 const float* scale, int64_t M, int64_t C, int64_t B, int res_dtype, int y_dtype,
                                 void* stream) {
    int st = res_check(n, M, C, B, res_dtype, y_dtype);
    if (st != MTLORA_OK) return st;
    if (!res || !y || !out) return MTLORA_ERR_NULL;
    ResParams p = {};
    for (int k = 0; k < n; ++k) {
        if (!res[k] || !y[k] || !out[k]) return MTLORA_ERR_NULL;
        if (((uintptr_t)res[k] | (uintptr_t)y[k] | (uintptr_t)out[k]) & 15u) return MTLORA_ERR_ALIGN;
        p.res[k] = res[k];
        p.y[k] = y[k];
        p.out[k] = out[k];
    }
    if (M == 0) return MTLORA_OK;
    p.scale = scale;
    p.M = M;
    p.C = (int)C;
    p.n = n;
    p.B = (int)B;
    p.rows_per_sample = M / B;
    hipStream_t s = (hipStream_t)stream;
    int64_t blocks = mtl_ceil_div(M * C / 8, 256);
    if (blocks > 2048) blocks = 2048;
    dim3 g((unsigned)blocks, (unsigned)n);
    const int er = mtl_elem_size(res_dtype), ey = mtl_elem_size(y_dtype);
    MtlProfScope prof(PK_RESIDUAL, (double)n * M * C * (2 * er + ey), s);
    if (res_dtype == MTLORA_F32 && y_dtype == MTLORA_F32)
        hipLaunchKernelGGL((k_residual_fwd<float, float>), g, dim3(256), 0, s, p);
    else if (res_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {
        if (res_dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_residual_fwd<float, f16>), g, dim3(256), 0, s, p);
        else if (y_dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_residual_fwd<f16, float>), g, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((k_residual_fwd<f16, f16>), g, dim3(256), 0, s, p);
    } else if (res_dtype == MTLORA_F32)
        hipLaunchKernelGGL((k_residual_fwd<float, bf16>), g, dim3(256), 0, s, p);
    else if (y_dtype == MTLORA_F32)
        hipLaunchKernelGGL((k_residual_fwd<bf16, float>), g, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((k_residual_fwd<bf16, bf16>), g, dim3(256), 0, s, p);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

/* g[k]: gradient of out_k (res dtype) or NULL; dy[k]: written (y dtype) where non-NULL; dres: sum_k g_k or NULL */
int mtlora_residual_droppath_bwd(int n, const void* const* g, void* const* dy, void* dres, const float* scale,
                                 int64_t M, int64_t C, int64_t B, int res_dtype, int y_dtype, void* stream) {
    int st = res_check(n, M, C, B, res_dtype, y_dtype);
    if (st != MTLORA_OK) return st;
    if (!g || !dy) return MTLORA_ERR_NULL;
    ResParams p = {};
    for (int k = 0; k < n; ++k) {
        if (((uintptr_t)g[k] | (uintptr_t)dy[k]) & 15u) return MTLORA_ERR_ALIGN;
        p.y[k] = g[k];
        p.out[k] = g[k] ? dy[k] : nullptr;
    }
    if (M == 0) return MTLORA_OK;
    p.dres = dres;
    p.scale = scale;
    p.M = M;
    p.C = (int)C;
    p.n = n;
    p.B = (int)B;
    p.rows_per_sample = M / B;
    hipStream_t s = (hipStream_t)stream;
    int64_t blocks = mtl_ceil_div(M * C / 8, 256);
    if (blocks > 2048) blocks = 2048;
    const int er = mtl_elem_size(res_dtype), ey = mtl_elem_size(y_dtype);
    MtlProfScope prof(PK_RESIDUAL, (double)M * C * (n * (er + ey) + (dres ? er : 0)), s);
    if (res_dtype == MTLORA_F32 && y_dtype == MTLORA_F32)
        hipLaunchKernelGGL((k_residual_bwd<float, float>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (res_dtype == MTLORA_F16 || y_dtype == MTLORA_F16) {
        if (res_dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_residual_bwd<float, f16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        else if (y_dtype == MTLORA_F32)
            hipLaunchKernelGGL((k_residual_bwd<f16, float>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((k_residual_bwd<f16, f16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else if (res_dtype == MTLORA_F32)
        hipLaunchKernelGGL((k_residual_bwd<float, bf16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (y_dtype == MTLORA_F32)
        hipLaunchKernelGGL((k_residual_bwd<bf16, float>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((k_residual_bwd<bf16, bf16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}
}
