// common.h -- shared device/host helpers for libmtlora_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtlora_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16;  // IEEE half: the reference's default autocast dtype (main.py:341); same MFMA rate as bf16 on gfx950
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MTL_WAVE 64

#define MTL_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return MTLORA_ERR_HIP;        \
    } while (0)

// Zero-fill as a KERNEL, not hipMemsetAsync: the buffers this library clears are consumed inside HIP graphs captured by
// the host framework, and a captured memset node was observed not to take effect on replay (stale pool memory showed
// through from the second replay on); kernel nodes replay reliably.  `bytes` must be a multiple of 4.
static __global__ void mtl_k_zero(uint32_t* p, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void mtl_zero_async(void* p, size_t bytes, hipStream_t s) {
    const size_t n = bytes / 4;
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mtl_k_zero, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), n);
}

__host__ __device__ static inline int64_t mtl_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t mtl_round_up(int64_t a, int64_t b) { return mtl_ceil_div(a, b) * b; }
static inline int mtl_elem_size(int dtype) { return dtype == MTLORA_F32 ? 4 : 2; }  // bf16 and f16: 2

// ---------------------------------------------------------------------------------------------
// counter-based dropout (restated in oracle/mtlora_oracle.py:dropout_keep_mask)
//   row hash   rh = mix32(m * 0x9E3779B1 + seed_lo + stream * 0x85EBCA77)
//   pair hash  h  = mix32(rh ^ ((k >> 1) + seed_hi * 0x27D4EB2F))
//   16 bits per element: keep <=> ((k & 1) ? h >> 16 : h & 0xFFFF) >= floor(p * 65536)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mtl_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

struct DropoutCfg {
    uint32_t seed_lo, seed_hi, thr16;  // thr16 == 0 -> disabled
    const unsigned long long* off;     // optional device word added to the seed at kernel start (graph replays)
    __host__ __device__ bool enabled() const { return thr16 != 0; }
};

static inline DropoutCfg mtl_make_dropout(float p, uint64_t seed, const uint64_t* seed_offset = nullptr) {
    DropoutCfg c;
    c.off = reinterpret_cast<const unsigned long long*>(seed_offset);
    c.seed_lo = (uint32_t)(seed & 0xFFFFFFFFu);
    c.seed_hi = (uint32_t)(seed >> 32);
    double t = (double)p * 65536.0;
    c.thr16 = p > 0.f ? (uint32_t)(t > 65535.0 ? 65535.0 : t) : 0u;
    return c;
}

// effective seed = seed + *off (mod 2^64), resolved once per kernel (one scalar load)
__device__ __forceinline__ void mtl_dropout_resolve(DropoutCfg& c) {
    if (c.off && c.thr16) {
        const unsigned long long e = (((unsigned long long)c.seed_hi << 32) | c.seed_lo) + *c.off;
        c.seed_lo = (uint32_t)e;
        c.seed_hi = (uint32_t)(e >> 32);
    }
    c.off = nullptr;
}
__device__ __forceinline__ uint32_t mtl_dropout_rowhash(const DropoutCfg& c, uint32_t stream, uint32_t m) {
    return mtl_mix32(m * 0x9E3779B1u + c.seed_lo + stream * 0x85EBCA77u);
}
// 32 bits covering elements (k & ~1, k | 1) of row m
__device__ __forceinline__ uint32_t mtl_dropout_pairbits(const DropoutCfg& c, uint32_t rowhash, uint32_t k) {
    return mtl_mix32(rowhash ^ ((k >> 1) + c.seed_hi * 0x27D4EB2Fu));
}
__device__ __forceinline__ bool mtl_dropout_keep(const DropoutCfg& c, uint32_t rowhash, uint32_t k) {
    uint32_t h = mtl_dropout_pairbits(c, rowhash, k);
    uint32_t bits = (k & 1u) ? (h >> 16) : (h & 0xFFFFu);
    return bits >= c.thr16;
}

// ---------------------------------------------------------------------------------------------
// element traits: one 16-byte vector = VEC elements
// ---------------------------------------------------------------------------------------------
template <typename T>
struct ET;
template <>
struct ET<float> {
    static constexpr int VEC = 4;
    static constexpr int DT = MTLORA_F32;
};
template <>
struct ET<bf16> {
    static constexpr int VEC = 8;
    static constexpr int DT = MTLORA_BF16;
};
template <>
struct ET<f16> {
    static constexpr int VEC = 8;
    static constexpr int DT = MTLORA_F16;
};

__device__ __forceinline__ float mtl_to_f32(float v) { return v; }
__device__ __forceinline__ float mtl_to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float mtl_to_f32(f16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T mtl_from_f32(float v);
template <>
__device__ __forceinline__ float mtl_from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16 mtl_from_f32<bf16>(float v) { return (bf16)v; }
template <>
__device__ __forceinline__ f16 mtl_from_f32<f16>(float v) { return (f16)v; }

// 16-byte vector view of VEC elements of T
template <typename T>
union Vec16 {
    u32x4 raw;
    T e[ET<T>::VEC];
};

template <typename T>
__device__ __forceinline__ Vec16<T> mtl_ld16(const T* p) {
    Vec16<T> v;
    v.raw = *reinterpret_cast<const u32x4*>(p);
    return v;
}
template <typename T>
__device__ __forceinline__ Vec16<T> mtl_zero16() {
    Vec16<T> v;
    v.raw = u32x4{0u, 0u, 0u, 0u};
    return v;
}

// register-only vector helpers (no unions: element access through a union can push the staging registers of a
// big kernel into scratch)
__device__ __forceinline__ uint32_t mtl_pack_bf16(float a, float b) {
    bf16 x = (bf16)a, y = (bf16)b;
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
__device__ __forceinline__ uint32_t mtl_pack_f16(float a, float b) {
    f16 x = (f16)a, y = (f16)b;  // v_cvt_f16_f32: round to nearest even
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
// the same packing as ONE instruction (a vector conversion: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round to nearest even, bit-identical
// results) for the wave-streaming kernels, whose epilogues are VALU-bound.  NOT used by the tiled kernels: with it the 8-wave
// masked-low-rank variants of k_nt (at the 128-VGPR cap) re-schedule into ~970 bytes of scratch per lane and run 8x slower.
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <typename T>
__device__ __forceinline__ uint32_t mtl_pk2(float a, float b) {
    if constexpr (__is_same(T, f16))
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
    else
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// two 16-bit elements of T in one dword <-> two floats (lo = element 0).  bf16: shifts; f16: conversions
template <typename T>
__device__ __forceinline__ uint32_t mtl_pack2(float a, float b) {
    if constexpr (__is_same(T, f16))
        return mtl_pack_f16(a, b);
    else
        return mtl_pack_bf16(a, b);
}
template <typename T>
__device__ __forceinline__ float mtl_lo2(uint32_t w) {
    if constexpr (__is_same(T, f16))
        return (float)__builtin_bit_cast(f16, (uint16_t)(w & 0xFFFFu));
    else
        return __builtin_bit_cast(float, w << 16);
}
template <typename T>
__device__ __forceinline__ float mtl_hi2(uint32_t w) {
    if constexpr (__is_same(T, f16))
        return (float)__builtin_bit_cast(f16, (uint16_t)(w >> 16));
    else
        return __builtin_bit_cast(float, w & 0xFFFF0000u);
}
// ---- erf GELU of 16-bit tensors without a transcendental --------------------------------------------------------------------------
// erf(x / sqrt 2) = z P(z^2) on |z| <= 3 (degree-8 minimax fit of erf(z) / z in z^2, fp32 Horner: |error| < 2.8e-5), saturated beyond
// (1 - erf(3) = 2.2e-5): GELU within 5.9e-5 absolute, GELU' within 1.4e-5 of the exact ones.  For results that are rounded to 16 bits
// (2^-9 relative) right after; fp32 tensors keep the Abramowitz-Stegun form (1.5e-7; one v_rcp + one v_exp per element, quarter rate:
// 25 issue slots per element against 8).  NP pairs at a time, step by step: the Horner steps of one pair depend on each other (and
// dependent packed operations cost a wait state each on gfx950), the pairs do not.
template <int NP>
__device__ __forceinline__ void mtl_erf2n(const f32x2 (&h)[NP], f32x2 (&er)[NP]) {
    f32x2 z[NP], s[NP], p[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        z[j] = h[j] * 0.70710678118654752f;
        z[j] = f32x2{__builtin_amdgcn_fmed3f(z[j].x, -3.f, 3.f), __builtin_amdgcn_fmed3f(z[j].y, -3.f, 3.f)};
        s[j] = z[j] * z[j];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) p[j] = 4.066549198e-08f * s[j] - 1.940831739e-06f;
    constexpr float cf[7] = {4.097715593e-05f, -5.101241795e-04f, 4.229743980e-03f, -2.508258229e-02f, 1.110399948e-01f,
                             -3.752788217e-01f, 1.128257636e+00f};
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int j = 0; j < NP; ++j) p[j] = p[j] * s[j] + cf[k];
#pragma unroll
    for (int j = 0; j < NP; ++j) er[j] = p[j] * z[j];
}
// the eight 16-bit values of a 16-byte vector: gelu(v), rounded once.  CVT: pack with the one-instruction conversion (mtl_pk2) or with
// the shift form (mtl_pack2) -- bit-identical results; the tiled kernels at their register cap schedule badly around the former
template <typename T, bool CVT>
__device__ __forceinline__ u32x4 mtl_gelu_pk4(const u32x4& v) {
    u32x4 o;
#pragma unroll
    for (int q2 = 0; q2 < 4; q2 += 2) {  // (two pairs in flight: four cost the streaming kernels ~20 registers and a wave per SIMD)
        f32x2 h[2], er[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) h[q] = f32x2{mtl_lo2<T>(v[q2 + q]), mtl_hi2<T>(v[q2 + q])};
        mtl_erf2n<2>(h, er);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 a = h[q] * (0.5f + 0.5f * er[q]);
            o[q2 + q] = CVT ? mtl_pk2<T>(a.x, a.y) : mtl_pack2<T>(a.x, a.y);
        }
    }
    return o;
}
// ... g .* gelu'(h): Phi(h) + h phi(h), the factor ATen's GeluBackward applies (one v_exp for the density)
template <typename T, bool CVT>
__device__ __forceinline__ u32x4 mtl_gelu_gate_pk4(const u32x4& g, const u32x4& hv) {
    u32x4 o;
#pragma unroll
    for (int q2 = 0; q2 < 4; q2 += 2) {
        f32x2 h[2], er[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) h[q] = f32x2{mtl_lo2<T>(hv[q2 + q]), mtl_hi2<T>(hv[q2 + q])};
        mtl_erf2n<2>(h, er);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 cdf = 0.5f + 0.5f * er[q];
            const f32x2 qq = h[q] * h[q];
            const f32x2 e = {__builtin_amdgcn_exp2f(qq.x * -0.72134752044448170f), __builtin_amdgcn_exp2f(qq.y * -0.72134752044448170f)};  // exp(-h^2 / 2)
            const f32x2 d = (cdf + h[q] * e * 0.39894228040143268f) * f32x2{mtl_lo2<T>(g[q2 + q]), mtl_hi2<T>(g[q2 + q])};
            o[q2 + q] = CVT ? mtl_pk2<T>(d.x, d.y) : mtl_pack2<T>(d.x, d.y);
        }
    }
    return o;
}

template <typename T>
struct VOps;
template <>
struct VOps<bf16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            f[2 * w] = __builtin_bit_cast(float, v[w] << 16);
            f[2 * w + 1] = __builtin_bit_cast(float, v[w] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
        return u32x4{mtl_pack_bf16(f[0], f[1]), mtl_pack_bf16(f[2], f[3]), mtl_pack_bf16(f[4], f[5]),
                     mtl_pack_bf16(f[6], f[7])};
    }
    // zero the dropped elements of the 8 consecutive elements starting at column k (k even)
    static __device__ __forceinline__ void drop(u32x4& v, const DropoutCfg& c, uint32_t rowhash, uint32_t k) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t h = mtl_dropout_pairbits(c, rowhash, k + 2 * w);
            uint32_t keep = 0u;
            if ((h & 0xFFFFu) >= c.thr16) keep |= 0x0000FFFFu;
            if ((h >> 16) >= c.thr16) keep |= 0xFFFF0000u;
            v[w] &= keep;
        }
    }
};
template <>
struct VOps<f16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            f[2 * w] = mtl_lo2<f16>(v[w]);
            f[2 * w + 1] = mtl_hi2<f16>(v[w]);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
        return u32x4{mtl_pack_f16(f[0], f[1]), mtl_pack_f16(f[2], f[3]), mtl_pack_f16(f[4], f[5]), mtl_pack_f16(f[6], f[7])};
    }
    static __device__ __forceinline__ void drop(u32x4& v, const DropoutCfg& c, uint32_t rowhash, uint32_t k) {
        VOps<bf16>::drop(v, c, rowhash, k);  // zeroing 16-bit lanes: the same bit masks
    }
};
template <>
struct VOps<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int w = 0; w < 4; ++w) f[w] = __builtin_bit_cast(float, v[w]);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
        return u32x4{__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                     __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3])};
    }
    static __device__ __forceinline__ void drop(u32x4& v, const DropoutCfg& c, uint32_t rowhash, uint32_t k) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t h = mtl_dropout_pairbits(c, rowhash, k + 2 * p);
            if ((h & 0xFFFFu) < c.thr16) v[2 * p] = 0u;
            if ((h >> 16) < c.thr16) v[2 * p + 1] = 0u;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// MFMA wrappers.  A "fragment" is the 2x16 bytes a lane holds for one 64-byte-wide k-tile:
// the SAME (lane, slot) -> k assignment is used for both operands, so the k order inside the
// instruction is irrelevant (dot products are permutation invariant); only the row/col <-> lane
// map (l & 31) and the C/D map matter:
//     D[i][j], lane l, reg r:  j = l & 31,  i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Frag {
    u32x4 v[2];
};

__device__ __forceinline__ void mtl_mma(const Frag<bf16>& a, const Frag<bf16>& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.v[0]), __builtin_bit_cast(bf16x8, b.v[0]),
                                                c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.v[1]), __builtin_bit_cast(bf16x8, b.v[1]),
                                                c, 0, 0, 0);
}
__device__ __forceinline__ void mtl_mma(const Frag<f16>& a, const Frag<f16>& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v[0]), __builtin_bit_cast(f16x8, b.v[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v[1]), __builtin_bit_cast(f16x8, b.v[1]), c, 0, 0, 0);
}
__device__ __forceinline__ void mtl_mma(const Frag<float>& a, const Frag<float>& b, f32x16& c) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 av = __builtin_bit_cast(f32x4, a.v[h]);
        f32x4 bv = __builtin_bit_cast(f32x4, b.v[h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], c, 0, 0, 0);
    }
}

// D-layout helpers for a 32x32 tile
__device__ __forceinline__ int mtl_d_row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int mtl_d_col(int lane) { return lane & 31; }

// ---------------------------------------------------------------------------------------------
// opt-in per-launch timing (mtlora_prof_begin / mtlora_prof_end): HIP events recorded on the launch
// stream around each kernel, tagged with a kind and the ALGORITHMIC bytes of that launch
// (SURVEY 8d formulas).  Inactive (one relaxed atomic load) unless enabled.
// ---------------------------------------------------------------------------------------------
enum MtlProfKind {
    PK_NT_FWD_MAIN = 0,  // k_nt: all 1+T outputs         reads X, writes Y_*        es*(MK + (1+T)MN)
    PK_NT_FWD_P = 1,     // k_nt: P = a D(X) A^T          reads x_t                  es*(T*x_t*MK)
    PK_NT_BWD_Q = 2,     // k_nt: Q = a dY B              (re-read of dY)            0
    PK_NT_BWD_DX = 3,    // k_nt: dX [+dX_t]              reads dY_*, writes dX_*    es*(n_dy*MN + (1+T*x_t)MK)
    PK_TN = 4,           // k_tn: dA / dB                 reads X, x_t               es*((1+T*x_t)MK)
    PK_ATTN_FWD = 5,     // k_attn_fwd                    qkv in, out                es*4*M*C
    PK_ATTN_BWD = 6,     // k_attn_bwd                    qkv, dout in, dqkv out     es*7*M*C
    PK_PACK = 7,
    PK_REDUCE = 8,
    PK_WINDOW = 9,
    PK_LN_FWD = 10,      // k_ln_fwd                      x in, y out
    PK_LN_BWD = 11,      // k_ln_bwd                      x, dy in, dx out
    PK_RESIDUAL = 13,    // k_residual_fwd / _bwd         residual + DropPath over 1+T tensors
    PK_BN = 12,          // k_bn_colsum / k_bn_apply      heads' BatchNorm(+ReLU)
    PK_LOSS = 14,        // k_up_loss                     low-res logits in, gradient out, labels in
    PK_SUM = 15,         // k_sum: G = sum of the output gradients (matrixv2 / pre-summed dX operand)
    PK_UPSAMPLE = 16,    // k_up_fwd / k_up_bwd: head's coarse -> fine bilinear maps
    PK_NT_PLAIN_FWD = 17,  // k_nt at rank 0: y = x W^T + b of the callers (heads, PatchMerging reduction, patch embed)
    PK_NT_PLAIN_DX = 18,   // k_nt at rank 0: dX = dY W of the same
    PK_TN_PLAIN = 19,      // k_tn through mtlora_gemm_tn: narrow-output weight gradients of the callers
    PK_COUNT = 24
};
int mtl_prof_start(int kind, double alg_bytes, hipStream_t s, double s8d_bytes = 0.0, double flops = 0.0);
void mtl_prof_tag(const char* fmt, ...);  // shape note attached to the NEXT record (MTLORA_PROF_DUMP=<file> lists records)
void mtl_prof_stop(int idx, hipStream_t s);
struct MtlProfScope {
    int idx;
    hipStream_t s;
    // bytes: useful bytes of the launch as issued; s8d: the SURVEY 8(d) share of them (hot-path kinds only); flops: algorithmic
    MtlProfScope(int kind, double bytes, hipStream_t st, double s8d = 0.0, double flops = 0.0)
        : idx(mtl_prof_start(kind, bytes, st, s8d, flops)), s(st) {}
    ~MtlProfScope() {
        if (idx >= 0) mtl_prof_stop(idx, s);
    }
};
