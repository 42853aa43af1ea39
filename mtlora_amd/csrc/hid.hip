// hid.h -- k_hid_*: the TASK streams of a task-enabled Mlp (fc1 -> GELU -> fc2 called with x_tasks, swin_transformer_mtlora.py:57-78 of the
// reference) without their 4C-wide tensors.  Included by linear.hip inside its anonymous namespace.
//
// In such an Mlp (reference lora.py:262-266 with x_tasks given) the hidden tensors of task t are
//     h_t = h_base + P1_t B1_t^T          h_base = xn W1^T + b1 (the SHARED input's pretrained product), P1_t = s_t xn_t A1_t^T  (M x r_t)
//     a_t = gelu(h_t)
// and fc2 reads a_t ONLY through its skinny projection P2_t = s_t a_t A2_t^T (M x r_t): y_t = a_s W2^T + b2 + P2_t B2_t^T takes its dense part from
// the SHARED activation.  Backward likewise: dH_t = (Q2_t A2_t) .* gelu'(h_t) is consumed by G = sum_o dH_o (the dense dX operand of fc1), by
// Q1_t = s_t dH_t B1_t (M x r_t) and by the two factor gradients dB1_t = dH_t^T P1_t, dA2_t = Q2_t^T a_t.  So h_t, a_t and dH_t -- 3 x T tensors of
// M x 4C elements, written once and read 2-3 times each by the per-layer path (k_nt MULTI, k_sp_proj, k_rank_out, k_sp_projsum, k_sp_tn) -- are
// functions of ONE M x 4C tensor (h_base) and a few M x r_t ones.  These kernels evaluate them in registers:
//     k_hid_proj   (forward)   P2_t = alpha2_t gelu(h_base + P1_t B1_t^T) A2_t^T                              reads h_base once for all tasks
//     k_hid_bwd    (backward)  G = dH_s + sum_t dH_t,  Q1_t,  per-workgroup partials of dB1_t / dA2_t          reads h_base, dH_s; writes G
//     k_hid_reduce             the fixed-order sum of those partials (deterministic: no float atomics)
// With r_t = 4 the per-element work is a handful of FMAs plus one erf: VALU work, not a GEMM (as k_rank_out).  A THREAD owns two adjacent hidden
// columns (its 2 x r_t factor values of B1_t and A2_t per task live in registers as float2: v_pk_fma_f32), the workgroup's hidden / 2 threads walk
// the rows together; the row-wise reductions (P2 / Q1: r_t values per task and row) go through a halving butterfly inside the wave (17 shuffles
// for 16 values) and a small double-buffered LDS table across the waves, one barrier per HID_RB rows.

#include "common.h"
#include "internal.h"
#include "hid_params.h"
#include <atomic>

namespace {

template <typename T>
__device__ __forceinline__ void sp_mma1(const u32x4& a, const u32x4& b, f32x16& c) {
    if constexpr (__is_same(T, f16))
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// exact (erf) GELU and its derivative with ONE erf / exp (Abramowitz-Stegun 7.1.26 as gelu_fwd / gelu_grad of linear.hip), two columns at a time
__device__ __forceinline__ void hid_gelu2(const f32x2 h, f32x2& a, f32x2& g) {
    const f32x2 z = f32x2{fabsf(h.x), fabsf(h.y)} * 0.70710678118654752f;
    const f32x2 d = 1.f + 0.3275911f * z;
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 zz = -(z * z);
    const f32x2 e = {__expf(zz.x), __expf(zz.y)};
    f32x2 p = 1.061405429f * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2 ea = 1.f - p * t * e;
    const f32x2 cdf = 0.5f + 0.5f * f32x2{copysignf(ea.x, h.x), copysignf(ea.y, h.y)};
    a = h * cdf;
    g = cdf + h * e * 0.39894228040143268f;
}
__device__ __forceinline__ f32x2 hid_gelu2_fwd(const f32x2 h) {
    const f32x2 z = f32x2{fabsf(h.x), fabsf(h.y)} * 0.70710678118654752f;
    const f32x2 d = 1.f + 0.3275911f * z;
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 zz = -(z * z);
    const f32x2 e = {__expf(zz.x), __expf(zz.y)};
    f32x2 p = 1.061405429f * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2 ea = 1.f - p * t * e;
    return h * (0.5f + 0.5f * f32x2{copysignf(ea.x, h.x), copysignf(ea.y, h.y)});
}

// RR consecutive elements of T at a WAVE-UNIFORM address -> floats (through SGPRs: the unpacking then runs on the scalar unit)
template <typename T, int RR>
__device__ __forceinline__ void hid_row_vals(const T* p, float (&f)[RR]) {
    static_assert(RR == 4 || RR == 8, "rank block");
    if constexpr (RR == 4) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(p);
        const uint32_t w0 = __builtin_amdgcn_readfirstlane(w[0]), w1 = __builtin_amdgcn_readfirstlane(w[1]);
        f[0] = mtl_lo2<T>(w0);
        f[1] = mtl_hi2<T>(w0);
        f[2] = mtl_lo2<T>(w1);
        f[3] = mtl_hi2<T>(w1);
    } else {
        const u32x4 w = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t wi = __builtin_amdgcn_readfirstlane(w[i]);
            f[2 * i] = mtl_lo2<T>(wi);
            f[2 * i + 1] = mtl_hi2<T>(wi);
        }
    }
}

// sum NV per-lane values over the 64 lanes: halving butterfly (lane keeps the upper half of the values when its mask bit is set and sends the
// other half to its partner), then an all-reduce of the single value left.  Returns the complete sum of value `idx` (a function of lane & (NV-1));
// lanes 0 .. NV-1 hold the NV distinct sums.  Fixed order: deterministic.
// m all-ones -> a, zero -> b, as ONE v_bfi_b32 on the VALUES (written as `up ? v[i] : v[half + i]` the compiler turns the select into a
// runtime INDEX into v[] and expands every access into a 16-way compare / select chain: 3 600 instructions per 4 rows)
__device__ __forceinline__ float hid_sel(uint32_t m, float a, float b) {
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, a) & m) | (__builtin_bit_cast(uint32_t, b) & ~m));
}
template <int NV>
__device__ __forceinline__ float hid_wave_reduce(float (&v)[NV], int lane, int& idx) {
    static_assert(NV == 1 || NV == 2 || NV == 4 || NV == 8 || NV == 16, "power of two");
    idx = 0;
#pragma unroll
    for (int half = NV / 2, mask = 1; half >= 1; half >>= 1, mask <<= 1) {
        const uint32_t up = (lane & mask) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float lo = v[i], hi = v[half + i];
            const float send = hid_sel(up, lo, hi);
            const float keep = hid_sel(up, hi, lo);
            v[i] = keep + __shfl_xor(send, mask);
        }
        idx = 2 * idx + (int)(up & 1u);
    }
    float x = v[0];
#pragma unroll
    for (int mask = NV; mask < 64; mask <<= 1) x += __shfl_xor(x, mask);
    return x;
}

// the workgroup's threads each hold the sums of their wave in lanes < NV; `fin(r, t)`-threads add the waves' entries
template <typename T, int TG, int RR>
__device__ __forceinline__ void hid_finish_rows(const float* red, int NW, int tid, int64_t row0, int64_t M, int nt, const float* alpha, const int* off,
                                                T* out, int ldo) {
    if (tid >= HID_RB * TG) return;
    const int r = tid / TG, t = tid - r * TG;
    const int64_t m = row0 + r;
    if (m >= M || t >= nt) return;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int rho = 0; rho < RR; ++rho) {
        float acc = 0.f;
        for (int w = 0; w < NW; ++w) acc += red[(r * HID_MAXW + w) * 16 + t * RR + rho];
        s[rho] = acc * alpha[off[t] + rho];
    }
    const u32x4 o = {mtl_pack2<T>(s[0], s[1]), mtl_pack2<T>(s[2], s[3]), mtl_pack2<T>(s[4], s[5]), mtl_pack2<T>(s[6], s[7])};
    *reinterpret_cast<u32x4*>(out + m * ldo + off[t]) = o;
}

template <typename T, int TG, int RR, int NTHR>
__global__ __launch_bounds__(NTHR) void k_hid_proj(const HidParams P) {
    constexpr int NV = TG * RR;
    static_assert(NV <= 16, "row values per launch");
    __shared__ float red[2][HID_RB * HID_MAXW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = (int)blockDim.x >> 6;
    const int H = P.H, c2 = 2 * tid, nt = P.nt;
    const T* hbase = reinterpret_cast<const T*>(P.hbase);
    const T* p1 = reinterpret_cast<const T*>(P.p1);
    f32x2 b1[TG][RR], a2[TG][RR];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int rho = 0; rho < RR; ++rho) {
            // (slots t >= nt carry task 0's offsets -- valid addresses -- and ZERO factors: their h_t = h_base, dH_t = 0, nothing of them is stored;
            // the row loop stays free of per-task branches)
            const float live = t < nt ? 1.f : 0.f;
            const uint32_t wb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.b1t) + (int64_t)(P.off1[t] + rho) * H + c2);
            const uint32_t wa = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.a2) + (int64_t)(P.off2[t] + rho) * H + c2);
            b1[t][rho] = f32x2{mtl_lo2<T>(wb), mtl_hi2<T>(wb)} * live;
            a2[t][rho] = f32x2{mtl_lo2<T>(wa), mtl_hi2<T>(wa)} * live;
        }
    const int64_t nblk = (P.M + HID_RB - 1) / HID_RB;
    int buf = 0;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x, buf ^= 1) {
        float* rd = red[buf];
        uint32_t hw[HID_RB];
#pragma unroll
        for (int r = 0; r < HID_RB; ++r) {
            const int64_t m = rb * HID_RB + r;
            hw[r] = m < P.M ? *reinterpret_cast<const uint32_t*>(hbase + m * H + c2) : 0u;
        }
#pragma unroll
        for (int r = 0; r < HID_RB; ++r) {
            const int64_t m = rb * HID_RB + r;
            if (m >= P.M) break;  // (uniform)
            const f32x2 hb = {mtl_lo2<T>(hw[r]), mtl_hi2<T>(hw[r])};
            float v[NV];
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                float pv[RR];
                hid_row_vals<T, RR>(p1 + m * P.ldp1 + P.off1[t], pv);
                f32x2 h = hb;
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) h += pv[rho] * b1[t][rho];
                const f32x2 a = hid_gelu2_fwd(h);
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    const f32x2 pr = a * a2[t][rho];
                    v[t * RR + rho] = pr.x + pr.y;
                }
            }
            int idx;
            const float s = hid_wave_reduce<NV>(v, lane, idx);
            if (lane < NV) rd[(r * HID_MAXW + wave) * 16 + idx] = s;
        }
        __syncthreads();
        hid_finish_rows<T, TG, RR>(rd, NW, tid, rb * HID_RB, P.M, nt, P.alpha2, P.off2, reinterpret_cast<T*>(P.p2), P.ldp2);
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA form of the forward kernel (round 6, second version).  The VALU form above is bound by its own instruction stream (~200 VALU per row
// and lane, 30 % of it the cross-lane butterfly) and by latency at <= 2-3 waves per SIMD: 1.0 ms at stage 0 of c2.  Here the row sums ARE a
// matrix product:  P2[m][(t, rho)] = sum_j a_t[m][j] A2_t[rho][j]  -- one v_mfma_f32_32x32x16 per task and 16-column step, the tasks of a launch
// sharing ONE accumulator (the B operand of task t is zero outside its own rank columns n = t RR + rho), no cross-lane reduction at all.
//   * lane (m = l & 31, jg = l >> 5) owns row m of the workgroup's 32-row block and, in step s of its wave, the 8 columns
//     w * 128 + 16 s + 8 jg ..: h_base arrives as one 16-byte load per step, the row's P1 values as one 16-byte load per task;
//   * h_t = h_base + sum_rho P1_t[m][rho] B1_t[:, rho]: the factor rows come from an fp32 LDS table (broadcast reads: a half-wave reads one
//     address), 4 v_pk_fma_f32 per rank column; gelu in registers; the 8 activations, packed to one 16-byte fragment, ARE the A operand;
//   * the B operand is a 16-byte LDS read of the A2 row (t, rho) = n (rows padded by 16 B: conflict-free), the zero row for the lanes of
//     other tasks; after the 8 steps the lanes n < nt RR hold the wave's share of P2 for 16 rows each: summed across the waves through LDS
//     (double-buffered: one barrier per row block), scaled by alpha2 and stored as whole 16-byte rank segments.
// ------------------------------------------------------------------------------------------------


__device__ __forceinline__ void hid_unpack8(const u32x4& v, f32x2 (&f)[4], bf16*) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = f32x2{__builtin_bit_cast(float, v[i] << 16), __builtin_bit_cast(float, v[i] & 0xFFFF0000u)};
}
__device__ __forceinline__ void hid_unpack8(const u32x4& v, f32x2 (&f)[4], f16*) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = f32x2{mtl_lo2<f16>(v[i]), mtl_hi2<f16>(v[i])};
}

template <typename T, int RR, int HC>
__global__ __launch_bounds__(HC / 2) void k_hid_proj_m(const HidParams P) {
    typedef HidMGeom<T, RR> G;
    constexpr int NV = G::NV;
    extern __shared__ __attribute__((aligned(16))) unsigned char hid_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int H = HC, NW = HC / HIDM_CW;  // (compile-time: every LDS table offset is an instruction immediate)
    const int nt = P.nt, m = lane & 31, jg = lane >> 5;
    constexpr int a2s = H * 2 + 16;
    float* b1f = reinterpret_cast<float*>(hid_smem);                               // [NV][H]
    unsigned char* a2b = hid_smem + (size_t)NV * H * 4;                            // [NV + 1][a2s]  (row NV: zeros)
    float* red = reinterpret_cast<float*>(a2b + (size_t)(NV + 1) * a2s);           // [2][NW][NV][32]
    // ---- factor tables (once per workgroup)
    for (int i = tid; i < NV * (H / 2); i += (int)blockDim.x) {
        const int row = i / (H / 2), c = 2 * (i - row * (H / 2)), t = row / RR, rho = row - t * RR;
        uint32_t wb = 0u, wa = 0u;
        if (t < nt) {
            wb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.b1t) + (int64_t)(P.off1[t] + rho) * H + c);
            wa = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.a2) + (int64_t)(P.off2[t] + rho) * H + c);
        }
        *reinterpret_cast<f32x2*>(b1f + (size_t)row * H + c) = f32x2{mtl_lo2<T>(wb), mtl_hi2<T>(wb)};
        *reinterpret_cast<uint32_t*>(a2b + (size_t)row * a2s + c * 2) = wa;
    }
    for (int i = tid; i < a2s / 4; i += (int)blockDim.x) *reinterpret_cast<uint32_t*>(a2b + (size_t)NV * a2s + i * 4) = 0u;
    __syncthreads();
    const T* hbase = reinterpret_cast<const T*>(P.hbase);
    const T* p1 = reinterpret_cast<const T*>(P.p1);
    const int col_w = wave * HIDM_CW + 8 * jg;   // this lane's first column of step 0
    const int n = lane & 31;
    // B-operand row of this lane per task: its own rank row, or the zero row
    int brow[HID_TG];
#pragma unroll
    for (int t = 0; t < HID_TG; ++t) brow[t] = ((n / RR) == t && n < NV ? n : NV) * a2s;
    const int64_t nblk = (P.M + 31) / 32;
    int buf = 0;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x, buf ^= 1) {
        const int64_t m0 = rb * 32;
        const int64_t row = (m0 + m < P.M) ? m0 + m : P.M - 1;  // (rows past M: a valid row, never stored)
        u32x4 hv[8], pw[HID_TG];
#pragma unroll
        for (int s = 0; s < 8; ++s) hv[s] = *reinterpret_cast<const u32x4*>(hbase + row * H + col_w + 16 * s);
#pragma unroll
        for (int t = 0; t < HID_TG; ++t) pw[t] = *reinterpret_cast<const u32x4*>(p1 + row * P.ldp1 + P.off1[t]);
        float pv[HID_TG][RR];
#pragma unroll
        for (int t = 0; t < HID_TG; ++t) {
            f32x2 q[4];
            hid_unpack8(pw[t], q, (T*)nullptr);
#pragma unroll
            for (int rho = 0; rho < RR; ++rho) pv[t][rho] = (rho & 1) ? q[rho >> 1].y : q[rho >> 1].x;
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int col = col_w + 16 * s;
            f32x2 hb[4];
            hid_unpack8(hv[s], hb, (T*)nullptr);
#pragma unroll
            for (int t = 0; t < HID_TG; ++t) {
                f32x2 h[4] = {hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    const float* br = b1f + (size_t)(t * RR + rho) * H + col;
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(br), b1v = *reinterpret_cast<const f32x4*>(br + 4);
                    const float pz = pv[t][rho];
                    h[0] += pz * f32x2{b0[0], b0[1]};
                    h[1] += pz * f32x2{b0[2], b0[3]};
                    h[2] += pz * f32x2{b1v[0], b1v[1]};
                    h[3] += pz * f32x2{b1v[2], b1v[3]};
                }
                u32x4 fa;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 a = hid_gelu2_fwd(h[i]);
                    fa[i] = mtl_pk2<T>(a.x, a.y);
                }
                const u32x4 fb = *reinterpret_cast<const u32x4*>(a2b + brow[t] + col * 2);
                sp_mma1<T>(fa, fb, acc);
            }
        }
        // the wave's share of P2: lane n holds rows (r & 3) + 8 (r >> 2) + 4 jg of column n
        float* rd = red + (size_t)buf * NW * NV * 32;
        if (n < NV) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rd[((size_t)wave * NV + n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * jg] = acc[r];
        }
        __syncthreads();
        if (tid < 32 * HID_TG) {
            const int rr = tid & 31, t = tid >> 5;
            if (t < nt && m0 + rr < P.M) {
                float sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[e] = 0.f;
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    float a = 0.f;
                    for (int w = 0; w < NW; ++w) a += rd[((size_t)w * NV + t * RR + rho) * 32 + rr];
                    sv[rho] = a * P.alpha2[P.off2[t] + rho];
                }
                const u32x4 o = {mtl_pack2<T>(sv[0], sv[1]), mtl_pack2<T>(sv[2], sv[3]), mtl_pack2<T>(sv[4], sv[5]), mtl_pack2<T>(sv[6], sv[7])};
                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(P.p2) + (m0 + rr) * P.ldp2 + P.off2[t]) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA form of the backward kernel.  Same lane map as k_hid_proj_m (lane = row m of a 32-row block x 8 columns per 16-column step), a wave owns
// HIDB_CW = 32 columns (two steps), the workgroup's H / 32 waves cover the row block's H columns together (H = 384: 12 waves, 3 per SIMD).
//   per (step, task):  h_t and u_t = Q2_t A2_t from two fp32 LDS tables (broadcast reads), gelu / gelu' in registers, dH_t = u_t gelu'(h_t),
//                      G += dH_t;  Q1 += dH_t B1_t^T as ONE mfma (dH_t packed = A operand; B = the B1 row of lane n, zero outside task t);
//                      the packed dH_t and a_t fragments also go to the wave's two [32 rows][32 columns] LDS images;
//   per task:          the two row reductions as mfma with BOTH operands read back transposed (ds_read_b64_tr_b16: a lane then holds 8
//                      consecutive rows of one column = the k axis):  dB1^T[j][n] += sum_m dH_t[m][j] P1[m][n],  dA2[j][n] += sum_m a_t[m][j] Q2[m][n],
//                      P1 / Q2 from workgroup-wide [32][32] images of the row block (columns n = t RR + rho), masked to task t's columns;
//   per row block:     the waves' Q1 shares meet in LDS (overlaying the images), alpha-scaled, stored as 16-byte rank segments;
//   at the end:        the two accumulators of the wave go to the workgroup's partial (k_hid_reduce sums the workgroups in fixed order).
// ------------------------------------------------------------------------------------------------


// transposed fragment of a [32 rows][>= col0 + 32 columns] 16-bit LDS image with 64-byte rows: lane (i = l & 31, hh = l >> 5) gets the 16
// elements Src[m][col0 + i], m in the slot order of tn_frag16 (linear.hip) -- the same order for both operands of an mfma pair
__device__ __forceinline__ void hid_tr_frag(const unsigned char* img, int col0, int lane, u32x4& f0, u32x4& f1) {
    const int g = lane >> 4, i = lane & 15, hh = g >> 1;
    const int col = col0 + 16 * (g & 1) + 4 * (i & 3);
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = ((j >> 1) * 16) + 8 * hh + 4 * (j & 1) + (i >> 2);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + row * 64 + col * 2));
        const u32x2 u = __builtin_bit_cast(u32x2, v);
        w[2 * j] = u[0];
        w[2 * j + 1] = u[1];
    }
    f0 = u32x4{w[0], w[1], w[2], w[3]};
    f1 = u32x4{w[4], w[5], w[6], w[7]};
}

template <typename T, int RR, int HC>
__global__ __launch_bounds__(HC * 2) void k_hid_bwd_m(const HidParams P) {
    typedef HidBGeom<T, RR> G_;
    constexpr int NV = G_::NV;
    extern __shared__ __attribute__((aligned(16))) unsigned char hid_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int H = HC, NW = HC / HIDB_CW;  // (compile-time: every LDS table offset is an instruction immediate)
    const int nt = P.nt, m = lane & 31, jg = lane >> 5, n = lane & 31;
    constexpr int bs = H * 2 + 16;
    float* b1f = reinterpret_cast<float*>(hid_smem);                                     // [NV][H]  B1 rows (h update)
    float* a2f = b1f + (size_t)NV * H;                                                   // [NV][H]  A2 rows (u)
    unsigned char* b1b = reinterpret_cast<unsigned char*>(a2f + (size_t)NV * H);         // [NV + 1][bs] 16-bit B1 rows (Q1's B operand), zero row
    unsigned char* pimg = b1b + (size_t)(NV + 1) * bs;                                   // [32][32] P1 columns n
    unsigned char* qimg = pimg + 32 * 64;                                                // [32][32] Q2 columns n
    unsigned char* tiles = qimg + 32 * 64;                                               // per wave: dH image, a image ([32][32] each)
    float* red = reinterpret_cast<float*>(tiles);                                        // overlays the tiles: [NW][NV][32]
    for (int i = tid; i < NV * (H / 2); i += (int)blockDim.x) {
        const int row = i / (H / 2), c = 2 * (i - row * (H / 2)), t = row / RR, rho = row - t * RR;
        uint32_t wb = 0u, wa = 0u;
        if (t < nt) {
            wb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.b1t) + (int64_t)(P.off1[t] + rho) * H + c);
            wa = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.a2) + (int64_t)(P.off2[t] + rho) * H + c);
        }
        *reinterpret_cast<f32x2*>(b1f + (size_t)row * H + c) = f32x2{mtl_lo2<T>(wb), mtl_hi2<T>(wb)};
        *reinterpret_cast<f32x2*>(a2f + (size_t)row * H + c) = f32x2{mtl_lo2<T>(wa), mtl_hi2<T>(wa)};
        *reinterpret_cast<uint32_t*>(b1b + (size_t)row * bs + c * 2) = wb;
    }
    for (int i = tid; i < bs / 4; i += (int)blockDim.x) *reinterpret_cast<uint32_t*>(b1b + (size_t)NV * bs + i * 4) = 0u;
    for (int i = tid; i < 2 * 32 * 64 / 4; i += (int)blockDim.x) *reinterpret_cast<uint32_t*>(pimg + i * 4) = 0u;  // (columns >= NV stay zero)
    __syncthreads();
    const T* hbase = reinterpret_cast<const T*>(P.hbase);
    const T* gsrc = reinterpret_cast<const T*>(P.gsrc);
    T* gout = reinterpret_cast<T*>(P.g);
    const T* p1 = reinterpret_cast<const T*>(P.p1);
    const T* q2 = reinterpret_cast<const T*>(P.q2);
    const int col_w = wave * HIDB_CW;          // the wave's first column
    unsigned char* dimg = tiles + (size_t)wave * (2 * 32 * 64);
    unsigned char* aimg = dimg + 32 * 64;
    int brow[HID_TG];
    uint32_t tmask[HID_TG];  // all-ones for the lanes whose column n belongs to task t
#pragma unroll
    for (int t = 0; t < HID_TG; ++t) {
        const bool own = (n / RR) == t && n < NV;
        brow[t] = (own ? n : NV) * bs;
        tmask[t] = own ? 0xFFFFFFFFu : 0u;
    }
    f32x16 accB, accA;
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[e] = accA[e] = 0.f;
    const int64_t nblk = (P.M + 31) / 32;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x) {
        const int64_t m0 = rb * 32;
        const bool live = m0 + m < P.M;
        const int64_t row = live ? m0 + m : P.M - 1;
        u32x4 hv[2], gv[2], pw[HID_TG], qw[HID_TG];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            hv[s2] = *reinterpret_cast<const u32x4*>(hbase + row * H + col_w + 16 * s2 + 8 * jg);
            gv[s2] = gsrc ? *reinterpret_cast<const u32x4*>(gsrc + row * H + col_w + 16 * s2 + 8 * jg) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int t = 0; t < HID_TG; ++t) {
            pw[t] = *reinterpret_cast<const u32x4*>(p1 + row * P.ldp1 + P.off1[t]);
            qw[t] = *reinterpret_cast<const u32x4*>(q2 + row * P.ldq2 + P.off2[t]);
            if (!live) pw[t] = qw[t] = u32x4{0u, 0u, 0u, 0u};  // rows past M add nothing to the factor gradients (dH = 0, Q2 = 0)
        }
        // the row block's P1 / Q2 columns n = t RR + rho as [32][32] images (wave 0: lanes jg = 0 write P1, jg = 1 write Q2)
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < HID_TG; ++t) {
                if (t < nt) {
                    unsigned char* dst = (jg ? qimg : pimg) + m * 64 + t * RR * 2;
                    const u32x4 v = jg ? qw[t] : pw[t];
                    if constexpr (RR == 4)
                        *reinterpret_cast<u32x2*>(dst) = u32x2{v[0], v[1]};
                    else
                        *reinterpret_cast<u32x4*>(dst) = v;
                }
            }
        }
        __syncthreads();  // images of this row block complete (and the previous block's `red` reads are done)
        f32x16 accQ;
#pragma unroll
        for (int e = 0; e < 16; ++e) accQ[e] = 0.f;
        f32x2 Gs[2][4];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) hid_unpack8(gv[s2], Gs[s2], (T*)nullptr);
#pragma unroll
        for (int t = 0; t < HID_TG; ++t) {
            float pv[RR], qv[RR];  // (unpacked per task: the raw words are what stays live across the task loop)
            {
                f32x2 a[4], b[4];
                hid_unpack8(pw[t], a, (T*)nullptr);
                hid_unpack8(qw[t], b, (T*)nullptr);
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    pv[rho] = (rho & 1) ? a[rho >> 1].y : a[rho >> 1].x;
                    qv[rho] = (rho & 1) ? b[rho >> 1].y : b[rho >> 1].x;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int col = col_w + 16 * s2 + 8 * jg;
                u32x4 fd, fa;
                // four columns at a time (two passes per 8-column fragment): half the live h / u / table registers of the 8-wide form
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    f32x2 h[2], u[2];
                    h[0] = f32x2{mtl_lo2<T>(hv[s2][2 * hf]), mtl_hi2<T>(hv[s2][2 * hf])};
                    h[1] = f32x2{mtl_lo2<T>(hv[s2][2 * hf + 1]), mtl_hi2<T>(hv[s2][2 * hf + 1])};
                    u[0] = u[1] = f32x2{0.f, 0.f};
#pragma unroll
                    for (int rho = 0; rho < RR; ++rho) {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b1f + (size_t)(t * RR + rho) * H + col + 4 * hf);
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a2f + (size_t)(t * RR + rho) * H + col + 4 * hf);
                        const float pz = pv[rho], qz = qv[rho];
                        h[0] += pz * f32x2{b0[0], b0[1]};
                        h[1] += pz * f32x2{b0[2], b0[3]};
                        u[0] += qz * f32x2{a0[0], a0[1]};
                        u[1] += qz * f32x2{a0[2], a0[3]};
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x2 av, gd;
                        hid_gelu2(h[i], av, gd);
                        const f32x2 dh = u[i] * gd;
                        Gs[s2][2 * hf + i] += dh;
                        fd[2 * hf + i] = mtl_pk2<T>(dh.x, dh.y);
                        fa[2 * hf + i] = mtl_pk2<T>(av.x, av.y);
                    }
                }
                const u32x4 fb = *reinterpret_cast<const u32x4*>(b1b + brow[t] + col * 2);
                sp_mma1<T>(fd, fb, accQ);
                *reinterpret_cast<u32x4*>(dimg + m * 64 + (16 * s2 + 8 * jg) * 2) = fd;
                *reinterpret_cast<u32x4*>(aimg + m * 64 + (16 * s2 + 8 * jg) * 2) = fa;
            }
            // row reductions of task t over this block's 32 rows (the images are private to the wave: in-order LDS, no barrier)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            u32x4 d0, d1, x0, x1, p0, p1f, q0, q1f;
            hid_tr_frag(dimg, 0, lane, d0, d1);
            hid_tr_frag(aimg, 0, lane, x0, x1);
            hid_tr_frag(pimg, 0, lane, p0, p1f);
            hid_tr_frag(qimg, 0, lane, q0, q1f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p0[i] &= tmask[t];
                p1f[i] &= tmask[t];
                q0[i] &= tmask[t];
                q1f[i] &= tmask[t];
            }
            sp_mma1<T>(d0, p0, accB);
            sp_mma1<T>(d1, p1f, accB);
            sp_mma1<T>(x0, q0, accA);
            sp_mma1<T>(x1, q1f, accA);
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4 o = {mtl_pk2<T>(Gs[s2][0].x, Gs[s2][0].y), mtl_pk2<T>(Gs[s2][1].x, Gs[s2][1].y), mtl_pk2<T>(Gs[s2][2].x, Gs[s2][2].y),
                             mtl_pk2<T>(Gs[s2][3].x, Gs[s2][3].y)};
            if (live) *reinterpret_cast<u32x4*>(gout + row * H + col_w + 16 * s2 + 8 * jg) = o;
        }
        __syncthreads();  // every wave is done with its images: `red` may overlay them
        if (n < NV) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((size_t)wave * NV + n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * jg] = accQ[r];
        }
        __syncthreads();
        if (tid < 32 * HID_TG) {
            const int rr = tid & 31, t = tid >> 5;
            if (t < nt && m0 + rr < P.M) {
                float sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[e] = 0.f;
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    float a = 0.f;
                    for (int w = 0; w < NW; ++w) a += red[((size_t)w * NV + t * RR + rho) * 32 + rr];
                    sv[rho] = a * P.alpha1[P.off1[t] + rho];
                }
                const u32x4 o = {mtl_pack2<T>(sv[0], sv[1]), mtl_pack2<T>(sv[2], sv[3]), mtl_pack2<T>(sv[4], sv[5]), mtl_pack2<T>(sv[6], sv[7])};
                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(P.q1) + (m0 + rr) * P.ldq1 + P.off1[t]) = o;
            }
        }
        // (the next block's first barrier orders these reads before the images are written again)
    }
    // the wave's share of the factor gradients: lane n holds rows a(r) (the wave's columns j = col_w + a) of column n
    if (n < nt * RR) {
        const int t = n / RR, rho = n - t * RR;
        float* part = P.part + (int64_t)blockIdx.x * nt * 2 * RR * H;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = col_w + (r & 3) + 8 * (r >> 2) + 4 * jg;
            part[((int64_t)(t * 2 + 0) * RR + rho) * H + j] = accB[r];
            part[((int64_t)(t * 2 + 1) * RR + rho) * H + j] = accA[r];
        }
    }
}

template <typename T, int TG, int RR, int NTHR>
__global__ __launch_bounds__(NTHR) void k_hid_bwd(const HidParams P) {
    constexpr int NV = TG * RR;
    static_assert(NV <= 16, "row values per launch");
    __shared__ float red[2][HID_RB * HID_MAXW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = (int)blockDim.x >> 6;
    const int H = P.H, c2 = 2 * tid, nt = P.nt;
    const T* hbase = reinterpret_cast<const T*>(P.hbase);
    const T* gsrc = reinterpret_cast<const T*>(P.gsrc);
    T* gout = reinterpret_cast<T*>(P.g);
    const T* p1 = reinterpret_cast<const T*>(P.p1);
    const T* q2 = reinterpret_cast<const T*>(P.q2);
    f32x2 b1[TG][RR], a2[TG][RR], accB[TG][RR], accA[TG][RR];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int rho = 0; rho < RR; ++rho) {
            const float live = t < nt ? 1.f : 0.f;  // (as k_hid_proj)
            accB[t][rho] = accA[t][rho] = f32x2{0.f, 0.f};
            const uint32_t wb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.b1t) + (int64_t)(P.off1[t] + rho) * H + c2);
            const uint32_t wa = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.a2) + (int64_t)(P.off2[t] + rho) * H + c2);
            b1[t][rho] = f32x2{mtl_lo2<T>(wb), mtl_hi2<T>(wb)} * live;
            a2[t][rho] = f32x2{mtl_lo2<T>(wa), mtl_hi2<T>(wa)} * live;
        }
    const int64_t nblk = (P.M + HID_RB - 1) / HID_RB;
    int buf = 0;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x, buf ^= 1) {
        float* rd = red[buf];
        uint32_t hw[HID_RB], gw[HID_RB];
#pragma unroll
        for (int r = 0; r < HID_RB; ++r) {
            const int64_t m = rb * HID_RB + r;
            hw[r] = m < P.M ? *reinterpret_cast<const uint32_t*>(hbase + m * H + c2) : 0u;
            gw[r] = (m < P.M && gsrc) ? *reinterpret_cast<const uint32_t*>(gsrc + m * H + c2) : 0u;
        }
#pragma unroll
        for (int r = 0; r < HID_RB; ++r) {
            const int64_t m = rb * HID_RB + r;
            if (m >= P.M) break;  // (uniform)
            const f32x2 hb = {mtl_lo2<T>(hw[r]), mtl_hi2<T>(hw[r])};
            f32x2 G = {mtl_lo2<T>(gw[r]), mtl_hi2<T>(gw[r])};
            float v[NV];
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                float pv[RR], qv[RR];
                hid_row_vals<T, RR>(p1 + m * P.ldp1 + P.off1[t], pv);
                hid_row_vals<T, RR>(q2 + m * P.ldq2 + P.off2[t], qv);
                f32x2 h = hb, u = {0.f, 0.f};
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    h += pv[rho] * b1[t][rho];
                    u += qv[rho] * a2[t][rho];
                }
                f32x2 a, gd;
                hid_gelu2(h, a, gd);
                const f32x2 dh = u * gd;
                G += dh;
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) {
                    const f32x2 pr = dh * b1[t][rho];
                    v[t * RR + rho] = pr.x + pr.y;
                    accB[t][rho] += dh * pv[rho];
                    accA[t][rho] += a * qv[rho];
                }
            }
            *reinterpret_cast<uint32_t*>(gout + m * H + c2) = mtl_pack2<T>(G.x, G.y);
            int idx;
            const float s = hid_wave_reduce<NV>(v, lane, idx);
            if (lane < NV) rd[(r * HID_MAXW + wave) * 16 + idx] = s;
        }
        __syncthreads();
        hid_finish_rows<T, TG, RR>(rd, NW, tid, rb * HID_RB, P.M, nt, P.alpha1, P.off1, reinterpret_cast<T*>(P.q1), P.ldq1);
    }
    // this workgroup's share of the factor gradients
    float* part = P.part + (int64_t)blockIdx.x * nt * 2 * RR * H;
#pragma unroll
    for (int t = 0; t < TG; ++t)
        if (t < nt) {
#pragma unroll
            for (int rho = 0; rho < RR; ++rho) {
                *reinterpret_cast<f32x2*>(part + ((int64_t)(t * 2 + 0) * RR + rho) * H + c2) = accB[t][rho];
                *reinterpret_cast<f32x2*>(part + ((int64_t)(t * 2 + 1) * RR + rho) * H + c2) = accA[t][rho];
            }
        }
}

__global__ __launch_bounds__(256) void k_hid_reduce(const HidRedParams P) {
    const int64_t per = (int64_t)P.nt * 2 * P.RR * P.H;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const int j = (int)(i % P.H);
    const int rho = (int)((i / P.H) % P.RR), kind = (int)((i / ((int64_t)P.H * P.RR)) & 1), t = (int)(i / ((int64_t)P.H * P.RR * 2));
    int r_t = 0;
    float* dst = nullptr;
#pragma unroll
    for (int q = 0; q < HID_TG; ++q)
        if (q == t) {
            r_t = P.r[q];
            dst = kind ? P.dA2[q] : P.dB1[q];
        }
    if (rho >= r_t || !dst) return;
    float s0 = 0.f, s1 = 0.f;
    int w = 0;
    for (; w + 1 < P.n_wg; w += 2) {
        s0 += P.part[(int64_t)w * per + i];
        s1 += P.part[(int64_t)(w + 1) * per + i];
    }
    if (w < P.n_wg) s0 += P.part[(int64_t)w * per + i];
    const float s = s0 + s1;
    if (kind)
        dst[(int64_t)rho * P.H + j] = s;
    else
        dst[(int64_t)j * r_t + rho] = s;
}

// ---- launcher (the host side that knows the layers' layouts lives in linear.hip)
bool hid_raise_lds(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if ((done.load(std::memory_order_relaxed) >> dev) & 1ull) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return true;
}
#define HID_RAISE_LDS(KERNEL)                                              \
    do {                                                                   \
        static std::atomic<unsigned long long> done__{0};                  \
        (void)hid_raise_lds(done__, (const void*)(KERNEL), 160 * 1024);    \
    } while (0)

template <typename T>
void hid_go(const HidLaunch& L, const HidParams& q, hipStream_t s) {
    const dim3 g((unsigned)L.n_wg), b((unsigned)L.nthr);
    switch (L.kind) {
        case 0:
#define HID_GO(K, TG_, RR_, NT_) hipLaunchKernelGGL((K<T, TG_, RR_, NT_>), g, b, 0, s, q)
            if (L.rr == 4 && L.tg == 4) HID_GO(k_hid_proj, 4, 4, 512);
            else if (L.rr == 4 && L.nthr <= 768) HID_GO(k_hid_proj, 2, 4, 768);
            else if (L.rr == 4) HID_GO(k_hid_proj, 2, 4, 1024);
            else if (L.tg == 2) HID_GO(k_hid_proj, 2, 8, 512);
            else if (L.nthr <= 768) HID_GO(k_hid_proj, 1, 8, 768);
            else HID_GO(k_hid_proj, 1, 8, 1024);
            break;
        case 1:
            if (L.rr == 4 && L.tg == 4) HID_GO(k_hid_bwd, 4, 4, 512);
            else if (L.rr == 4 && L.nthr <= 768) HID_GO(k_hid_bwd, 2, 4, 768);
            else if (L.rr == 4) HID_GO(k_hid_bwd, 2, 4, 1024);
            else if (L.tg == 2) HID_GO(k_hid_bwd, 2, 8, 512);
            else if (L.nthr <= 768) HID_GO(k_hid_bwd, 1, 8, 768);
            else HID_GO(k_hid_bwd, 1, 8, 1024);
#undef HID_GO
            break;
        case 2:
#define HID_M(K, RR_, HC_)                                                 \
    do {                                                                   \
        HID_RAISE_LDS((K<T, RR_, HC_>));                                   \
        hipLaunchKernelGGL((K<T, RR_, HC_>), g, b, L.lds, s, q);           \
    } while (0)
            if (L.rr == 4 && q.H == 384) HID_M(k_hid_proj_m, 4, 384);
            else if (L.rr == 4 && q.H == 768) HID_M(k_hid_proj_m, 4, 768);
            else if (L.rr == 4 && q.H == 512) HID_M(k_hid_proj_m, 4, 512);
            else if (L.rr == 4 && q.H == 1024) HID_M(k_hid_proj_m, 4, 1024);
            else if (q.H == 384) HID_M(k_hid_proj_m, 8, 384);
            else if (q.H == 768) HID_M(k_hid_proj_m, 8, 768);
            break;
        case 3:
            if (L.rr == 4 && q.H == 384) HID_M(k_hid_bwd_m, 4, 384);
            else if (q.H == 384) HID_M(k_hid_bwd_m, 8, 384);
#undef HID_M
            break;
        default: break;
    }
}
}  // namespace

MTL_INTERNAL void mtli_hid_launch(const HidLaunch* L, const HidParams* q, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (L->dtype == MTLORA_F16)
        hid_go<f16>(*L, *q, s);
    else
        hid_go<bf16>(*L, *q, s);
}
MTL_INTERNAL void mtli_hid_reduce(const HidRedParams* r, int64_t per, void* stream) {
    hipLaunchKernelGGL(k_hid_reduce, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *r);
}
