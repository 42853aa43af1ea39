// hid.hip -- k_hid_*: the TASK streams of a task-enabled Mlp (fc1 -> GELU -> fc2 called with x_tasks, swin_transformer_mtlora.py:57-78 of the
// reference) without their 4C-wide tensors.  Its own translation unit; launched by linear.hip through internal.h (mtli_hid_launch / mtli_hid_rows_finish / mtli_hid_reduce).
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
#include <type_traits>

namespace {

template <typename T>
__device__ __forceinline__ void sp_mma1(const u32x4& a, const u32x4& b, f32x16& c) {
    if constexpr (__is_same(T, f16))
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// erf GELU through the transcendental-free erf of common.h (mtl_erf2n): these kernels are VALU-bound (SQ_ACTIVE_INST_VALU 73-77 % of the busy
// cycles, profiles/r06_pmc_hid.txt) and the Abramowitz-Stegun form was 43 % of their issue slots
__device__ __forceinline__ f32x2 hid_erf2(const f32x2 h) {
    const f32x2 hh[1] = {h};
    f32x2 er[1];
    mtl_erf2n<1>(hh, er);
    return er[0];
}
template <int NP>
__device__ __forceinline__ void hid_gelu2n_fwd(const f32x2 (&h)[NP], f32x2 (&a)[NP]) {
    f32x2 er[NP];
    mtl_erf2n<NP>(h, er);
#pragma unroll
    for (int j = 0; j < NP; ++j) a[j] = h[j] * (0.5f + 0.5f * er[j]);
}
// erf GELU and its derivative (one v_exp for the density term), two columns at a time
__device__ __forceinline__ void hid_gelu2(const f32x2 h, f32x2& a, f32x2& g) {
    const f32x2 cdf = 0.5f + 0.5f * hid_erf2(h);
    const f32x2 q = h * h;
    const f32x2 e = {__builtin_amdgcn_exp2f(q.x * -0.72134752044448170f), __builtin_amdgcn_exp2f(q.y * -0.72134752044448170f)};  // exp(-h^2 / 2)
    a = h * cdf;
    g = cdf + h * e * 0.39894228040143268f;
}
__device__ __forceinline__ f32x2 hid_gelu2_fwd(const f32x2 h) { return h * (0.5f + 0.5f * hid_erf2(h)); }

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
// MFMA forms (k_hid_fwd_d / k_hid_bwd_d; rank <= 4, 4 tasks per launch: n = 4 t + rho).  The VALU forms above are bound by their own instruction
// stream (~200 / ~310 VALU per row and lane, a third of it cross-lane butterflies) at 2-3 waves per SIMD: 1.0 / 2.6 ms at stage 0 of c2.
// Here every contraction is a v_mfma_f32_32x32x16 and the element-wise work happens in the ACCUMULATOR layout:
//   * a workgroup owns 32-row blocks x one chunk of HC hidden columns (blockIdx.y), wave w the 32 columns cw = 32 w of it;
//     "D layout" = lane (c = l & 31, hh = l >> 5) holds rows a(r) = (r & 3) + 8 (r >> 2) + 4 hh, r = 0..15, of column cw + c;
//   * h_t = h_base + P1_t B1_t^T is ONE mfma: A = the row block's P1 values (lane (m, kg): columns n = 8 kg .. 8 kg + 7 of a [32][32] LDS image),
//     B = task t's four B1 values of column c placed in its own k slots (zero elsewhere; 8 bytes per (t, c) in LDS), C = h_base in D layout;
//     u_t = Q2_t A2_t likewise (backward).  No per-element FMAs, no factor table traffic.
//   * gelu / gelu' on the 16 accumulator values; dH_t = u_t gelu'(h_t); G += dH_t in registers;
//   * the ROW reductions dB1_t^T[c][n] = sum_m dH_t[m][c] P1[m][n], dA2_t[c][n] = sum_m a_t[m][c] Q2[m][n] take their A operand straight from the
//     registers (a lane's 16 values ARE 16 rows of one column = the k axis) and their B operand from the P1 / Q2 images by transposed reads
//     (ds_read_b64_tr_b16) in the same row order, masked to task t's columns: all tasks share one accumulator each;
//   * the COLUMN reduction (forward P2 = a_t A2_t^T, backward Q1 = dH_t B1_t^T) needs rows along lanes: the packed values go through a per-wave
//     [32][32] LDS image (4 x 8-byte writes, read back transposed) once per task;
//   * h_base / dH_s enter and G leaves through the same image (row-major global accesses of 16 bytes per lane);
//   * the waves' shares of the row sums meet in LDS and go, unscaled fp32, to a (chunk, row) partial that k_hid_rows_finish adds over the chunks,
//     scales by alpha and stores as 16-byte rank segments; the factor-gradient accumulators go to the workgroup's partial (k_hid_reduce).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint32_t hid_pk(float a, float b) { return mtl_pk2<T>(a, b); }

// LDS images: [32 rows][32 cols] of 16-bit elements, 64-byte rows, the four 16-byte chunks of a row XOR-permuted by (row >> 2) & 3.
// Why: the transposed reads want a row stride of 16 dwords (mod 64) -- then the 4 rows x 64 bytes a ds_read_b64_tr_b16 cycle touches cover
// all 64 banks -- but with that stride every ROW-MAJOR access (lane = row: the A operands, the h_base / G transposes) puts rows r, r + 4,
// r + 8 ... on the same banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.70 (backward) / 0.75 (forward), profiles/r06_pmc_hid.txt.  The
// permutation is constant over an aligned group of 4 rows, so a transposed read still sees whole rows, and it spreads the eight rows
// that share a bank group over the four chunk positions: 2 lanes per bank = the minimum for 512 bytes.
__device__ __forceinline__ int hid_off16(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ int hid_off8(int row, int e) {  // element offset e (a multiple of 4) of the row
    return row * 64 + (((e >> 3) ^ ((row >> 2) & 3)) << 4) + (((e >> 2) & 1) << 3);
}
// image -> lane (c, hh): the 16 rows a(r) of column c, raw (two fragments of 8) or as floats
__device__ __forceinline__ void hid_dl_raw(const unsigned char* img, int lane, u32x4& f0, u32x4& f1) {
    const int g = lane >> 4, i = lane & 15, hh = g >> 1;
    const int col = 16 * (g & 1) + 4 * (i & 3);
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * q + 4 * hh + (i >> 2);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + hid_off8(row, col)));
        const u32x2 u = __builtin_bit_cast(u32x2, v);
        w[2 * q] = u[0];
        w[2 * q + 1] = u[1];
    }
    f0 = u32x4{w[0], w[1], w[2], w[3]};
    f1 = u32x4{w[4], w[5], w[6], w[7]};
}
template <typename T>
__device__ __forceinline__ void hid_dl_f32(const unsigned char* img, int lane, f32x16& v) {
    u32x4 f0, f1;
    hid_dl_raw(img, lane, f0, f1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = mtl_lo2<T>(f0[i]);
        v[2 * i + 1] = mtl_hi2<T>(f0[i]);
        v[8 + 2 * i] = mtl_lo2<T>(f1[i]);
        v[8 + 2 * i + 1] = mtl_hi2<T>(f1[i]);
    }
}
// lane (c, hh): its 16 packed D-layout values (f0 = rows r 0..7, f1 = r 8..15) -> image row c, i.e. the TRANSPOSE [32 cols][32 rows]
__device__ __forceinline__ void hid_dl_store_t(unsigned char* img, int lane, const u32x4& f0, const u32x4& f1) {
    const int c = lane & 31, hh = lane >> 5;
    *reinterpret_cast<u32x2*>(img + hid_off8(c, 4 * hh)) = u32x2{f0[0], f0[1]};        // rows 4 hh + 0..3
    *reinterpret_cast<u32x2*>(img + hid_off8(c, 8 + 4 * hh)) = u32x2{f0[2], f0[3]};    // rows 8 + 4 hh ..
    *reinterpret_cast<u32x2*>(img + hid_off8(c, 16 + 4 * hh)) = u32x2{f1[0], f1[1]};   // rows 16 + 4 hh ..
    *reinterpret_cast<u32x2*>(img + hid_off8(c, 24 + 4 * hh)) = u32x2{f1[2], f1[3]};   // rows 24 + 4 hh ..
}
// transposed fragment pair: lane (i = l & 31, hh) gets Img[8 hh + e][i] (f0) and Img[16 + 8 hh + e][i] (f1)
__device__ __forceinline__ void hid_tr_frag(const unsigned char* img, int lane, u32x4& f0, u32x4& f1) {
    const int g = lane >> 4, i = lane & 15, hh = g >> 1;
    const int col = 16 * (g & 1) + 4 * (i & 3);
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = ((j >> 1) * 16) + 8 * hh + 4 * (j & 1) + (i >> 2);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + hid_off8(row, col)));
        const u32x2 u = __builtin_bit_cast(u32x2, v);
        w[2 * j] = u[0];
        w[2 * j + 1] = u[1];
    }
    f0 = u32x4{w[0], w[1], w[2], w[3]};
    f1 = u32x4{w[4], w[5], w[6], w[7]};
}
// (the explicit waits are not what the kernels wait for: compiled out, forward 0.337 -> 0.327 ms, backward 0.559 -> 0.558 at stage 0)
#define HID_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)

// task t's four factor values of a column (8 bytes) as the B operand of the rank update: k slot e of lane (c, kg) is n = 8 kg + e
template <int t>
__device__ __forceinline__ u32x4 hid_place(const u32x2 tb, int kg) {
    const uint32_t sel = kg == (t >> 1) ? 0xFFFFFFFFu : 0u;
    if constexpr (t & 1)
        return u32x4{0u, 0u, tb[0] & sel, tb[1] & sel};
    else
        return u32x4{tb[0] & sel, tb[1] & sel, 0u, 0u};
}

// shared prologue: compact factor tables + projection rows of this chunk
template <typename T, int HC, bool BWD>
__device__ __forceinline__ void hid_d_tables(const HidParams& P, int tid, int nthr, int c_base, unsigned char* tabH, unsigned char* tabU, unsigned char* bq) {
    const int nt = P.nt, H = P.H;
    constexpr int bs = HC * 2 + 16;
    const uint16_t* b1 = reinterpret_cast<const uint16_t*>(P.b1t);
    const uint16_t* a2 = reinterpret_cast<const uint16_t*>(P.a2);
    for (int i = tid; i < HID_TG * HC; i += nthr) {
        const int t = i / HC, cc = i - t * HC;
        uint32_t h0 = 0u, h1 = 0u, u0 = 0u, u1 = 0u;
        if (t < nt) {
            const int64_t cb = c_base + cc;
            h0 = (uint32_t)b1[(int64_t)(P.off1[t] + 0) * H + cb] | ((uint32_t)b1[(int64_t)(P.off1[t] + 1) * H + cb] << 16);
            h1 = (uint32_t)b1[(int64_t)(P.off1[t] + 2) * H + cb] | ((uint32_t)b1[(int64_t)(P.off1[t] + 3) * H + cb] << 16);
            if (BWD) {
                u0 = (uint32_t)a2[(int64_t)(P.off2[t] + 0) * H + cb] | ((uint32_t)a2[(int64_t)(P.off2[t] + 1) * H + cb] << 16);
                u1 = (uint32_t)a2[(int64_t)(P.off2[t] + 2) * H + cb] | ((uint32_t)a2[(int64_t)(P.off2[t] + 3) * H + cb] << 16);
            }
        }
        *reinterpret_cast<u32x2*>(tabH + (size_t)i * 8) = u32x2{h0, h1};
        if (BWD) *reinterpret_cast<u32x2*>(tabU + (size_t)i * 8) = u32x2{u0, u1};
    }
    // projection rows n = 4 t + rho: forward A2 rows (P2 = a A2^T), backward B1 rows (Q1 = dH B1^T); row 16 = zeros
    const uint16_t* src = BWD ? b1 : a2;
    for (int i = tid; i < 16 * (HC / 2); i += nthr) {
        const int row = i / (HC / 2), c2 = 2 * (i - row * (HC / 2)), t = row >> 2, rho = row & 3;
        uint32_t w = 0u;
        if (t < nt) w = *reinterpret_cast<const uint32_t*>(src + (int64_t)((BWD ? P.off1[t] : P.off2[t]) + rho) * H + c_base + c2);
        *reinterpret_cast<uint32_t*>(bq + (size_t)row * bs + c2 * 2) = w;
    }
    for (int i = tid; i < bs / 4; i += nthr) *reinterpret_cast<uint32_t*>(bq + (size_t)16 * bs + i * 4) = 0u;
}

template <typename T, int HC>
__global__ __launch_bounds__(HC * 2, HC == 384 ? 3 : (HC == 128 ? 4 : 2)) void k_hid_fwd_d(const HidParams P) {
    constexpr int NW = HC / 32, bs = HC * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char hid_smem[];
    unsigned char* tabH = hid_smem;                              // [4][HC] x 8 B
    unsigned char* bq = tabH + (size_t)HID_TG * HC * 8;           // [17][bs]
    unsigned char* pimg = bq + (size_t)17 * bs;                   // [32][32]
    unsigned char* wimgs = pimg + 2048;                           // [NW][32][32]; `red` overlays it
    float* red = reinterpret_cast<float*>(wimgs);                 // [NW][16][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = P.nt, H = P.H, m = lane & 31, kg = lane >> 5, c = lane & 31, hh = lane >> 5, n = lane & 31;
    const int chunk = blockIdx.y, c_base = chunk * HC, cw = wave * 32;
    hid_d_tables<T, HC, false>(P, tid, NW * 64, c_base, tabH, nullptr, bq);
    for (int i = tid; i < 2048 / 4; i += NW * 64) *reinterpret_cast<uint32_t*>(pimg + i * 4) = 0u;
    __syncthreads();
    unsigned char* wimg = wimgs + (size_t)wave * 2048;
    const T* hbase = reinterpret_cast<const T*>(P.hbase);
    const T* p1 = reinterpret_cast<const T*>(P.p1);
    int brow[HID_TG];
#pragma unroll
    for (int t = 0; t < HID_TG; ++t) brow[t] = ((n >> 2) == t && n < 16 ? n : 16) * bs;
    const int64_t nblk = (P.M + 31) / 32;
    // the next row block's h_base (and, in wave 0, P1) words are requested while the current block is computed: the three barriers of a
    // block keep the waves of a workgroup in step, so nothing else would cover that latency
    u32x4 nh0 = {0u, 0u, 0u, 0u}, nh1 = nh0;
    u32x2 np[2] = {{0u, 0u}, {0u, 0u}};
    auto fetch = [&](int64_t rb) __attribute__((always_inline)) {
        if (rb >= nblk) return;
        const int64_t m0 = rb * 32;
        const bool live = m0 + m < P.M;
        const int64_t row = live ? m0 + m : P.M - 1;
        nh0 = *reinterpret_cast<const u32x4*>(hbase + row * H + c_base + cw + 16 * kg);
        nh1 = *reinterpret_cast<const u32x4*>(hbase + row * H + c_base + cw + 16 * kg + 8);
        if (wave == 0) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * kg + tt;
                np[tt] = u32x2{0u, 0u};
                if (t < nt && live) np[tt] = *reinterpret_cast<const u32x2*>(p1 + row * P.ldp1 + P.off1[t]);
            }
        }
    };
    fetch(blockIdx.x);
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x) {
        const int64_t m0 = rb * 32;
        const u32x4 hv0 = nh0, hv1 = nh1;
        if (wave == 0) {  // the row block's P1 columns n = 4 t + rho: lanes kg = 0 write tasks 0, 1, lanes kg = 1 tasks 2, 3
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) *reinterpret_cast<u32x2*>(pimg + hid_off8(m, (2 * kg + tt) * 4)) = np[tt];
        }
        __syncthreads();  // pimg complete; the previous block's `red` reads are done
        fetch(rb + gridDim.x);
        *reinterpret_cast<u32x4*>(wimg + hid_off16(m, 2 * kg)) = hv0;
        *reinterpret_cast<u32x4*>(wimg + hid_off16(m, 2 * kg + 1)) = hv1;
        HID_LGKM0();
        __builtin_amdgcn_wave_barrier();
        f32x16 hbD;
        hid_dl_f32<T>(wimg, lane, hbD);
        const u32x4 fp = *reinterpret_cast<const u32x4*>(pimg + hid_off16(m, kg));
        HID_LGKM0();
        __builtin_amdgcn_wave_barrier();
        f32x16 accP;
#pragma unroll
        for (int e = 0; e < 16; ++e) accP[e] = 0.f;
        auto task = [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            const u32x2 tb = *reinterpret_cast<const u32x2*>(tabH + ((size_t)t * HC + cw + c) * 8);
            f32x16 h = hbD;
            sp_mma1<T>(fp, hid_place<t>(tb, kg), h);
            u32x4 a0, a1;
#pragma unroll
            for (int i4 = 0; i4 < 2; ++i4) {  // four pairs at a time (independent Horner chains)
                f32x2 hv[4], av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = f32x2{h[8 * i4 + 2 * j], h[8 * i4 + 2 * j + 1]};
                hid_gelu2n_fwd<4>(hv, av);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = hid_pk<T>(av[j].x, av[j].y);
                    if (i4 == 0)
                        a0[j] = w;
                    else
                        a1[j] = w;
                }
            }
            hid_dl_store_t(wimg, lane, a0, a1);
            HID_LGKM0();
            __builtin_amdgcn_wave_barrier();
            u32x4 f0, f1;
            hid_tr_frag(wimg, lane, f0, f1);
            const u32x4 fb0 = *reinterpret_cast<const u32x4*>(bq + brow[t] + (cw + 8 * hh) * 2);
            const u32x4 fb1 = *reinterpret_cast<const u32x4*>(bq + brow[t] + (cw + 16 + 8 * hh) * 2);
            sp_mma1<T>(f0, fb0, accP);
            sp_mma1<T>(f1, fb1, accP);
            __builtin_amdgcn_wave_barrier();
        };
        task(std::integral_constant<int, 0>{});
        task(std::integral_constant<int, 1>{});
        task(std::integral_constant<int, 2>{});
        task(std::integral_constant<int, 3>{});
        __syncthreads();  // every wave is done with its image: `red` may overlay them
        if (n < 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((size_t)wave * 16 + n) * 33 + (r & 3) + 8 * (r >> 2) + 4 * hh] = accP[r];
        }
        __syncthreads();
        for (int f = tid; f < 512; f += NW * 64) {
            const int rr = f & 31, nn = f >> 5;
            if (m0 + rr < P.M) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) a += red[((size_t)w * 16 + nn) * 33 + rr];
                P.rowpart[((int64_t)chunk * P.M + m0 + rr) * 16 + nn] = a;
            }
        }
    }
}

template <typename T, int HC>
__global__ __launch_bounds__(HC * 2, HC == 384 ? 3 : (HC == 128 ? 3 : 2)) void k_hid_bwd_d(const HidParams P) {
    constexpr int NW = HC / 32, bs = HC * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char hid_smem[];
    unsigned char* tabH = hid_smem;                              // [4][HC] x 8 B   B1
    unsigned char* tabU = tabH + (size_t)HID_TG * HC * 8;         // [4][HC] x 8 B   A2
    unsigned char* bq = tabU + (size_t)HID_TG * HC * 8;           // [17][bs]        B1 rows
    unsigned char* pimg = bq + (size_t)17 * bs;                   // [32][32]: columns n = 4 t + rho: P1, columns 16 + n: Q2
    unsigned char* wimgs = pimg + 4096;                           // [NW][32][32]; `red` overlays it  (4096: layout kept, second half unused)
    float* red = reinterpret_cast<float*>(wimgs);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = P.nt, H = P.H, m = lane & 31, kg = lane >> 5, c = lane & 31, hh = lane >> 5, n = lane & 31;
    const int chunk = blockIdx.y, c_base = chunk * HC, cw = wave * 32;
    hid_d_tables<T, HC, true>(P, tid, NW * 64, c_base, tabH, tabU, bq);
    for (int i = tid; i < 4096 / 4; i += NW * 64) *reinterpret_cast<uint32_t*>(pimg + i * 4) = 0u;
    __syncthreads();
    unsigned char* wimg = wimgs + (size_t)wave * 2048;
    const T* hbase = reinterpret_cast<const T*>(P.hbase);
    const T* gsrc = reinterpret_cast<const T*>(P.gsrc);
    T* gout = reinterpret_cast<T*>(P.g);
    const T* p1 = reinterpret_cast<const T*>(P.p1);
    const T* q2 = reinterpret_cast<const T*>(P.q2);
    f32x16 accF;  // factor gradients of the wave's 32 columns: [column][n]: n < 16 dB1^T, n >= 16 dA2
#pragma unroll
    for (int e = 0; e < 16; ++e) accF[e] = 0.f;
    const int64_t nblk = (P.M + 31) / 32;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x) {
        const int64_t m0 = rb * 32;
        const bool live = m0 + m < P.M;
        const int64_t row = live ? m0 + m : P.M - 1;
        const int64_t go = row * H + c_base + cw + 16 * kg;
        const u32x4 hv0 = *reinterpret_cast<const u32x4*>(hbase + go), hv1 = *reinterpret_cast<const u32x4*>(hbase + go + 8);
        u32x4 gv0 = {0u, 0u, 0u, 0u}, gv1 = {0u, 0u, 0u, 0u};
        if (gsrc) {
            gv0 = *reinterpret_cast<const u32x4*>(gsrc + go);
            gv1 = *reinterpret_cast<const u32x4*>(gsrc + go + 8);
        }
        if (wave < 2) {  // wave 0: P1 image, wave 1: Q2 image (rows past M: zeros -> nothing reaches the factor gradients)
            const T* src = wave ? q2 : p1;
            const int64_t ld = wave ? P.ldq2 : P.ldp1;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * kg + tt;
                u32x2 v = {0u, 0u};
                if (t < nt && live) v = *reinterpret_cast<const u32x2*>(src + row * ld + (wave ? P.off2[t] : P.off1[t]));
                *reinterpret_cast<u32x2*>(pimg + hid_off8(m, 16 * wave + t * 4)) = v;
            }
        }
        __syncthreads();
        u32x4 hb0, hb1;  // h_base in D layout, packed (unpacked into the accumulator of each task's rank update)
        *reinterpret_cast<u32x4*>(wimg + hid_off16(m, 2 * kg)) = hv0;
        *reinterpret_cast<u32x4*>(wimg + hid_off16(m, 2 * kg + 1)) = hv1;
        HID_LGKM0();
        __builtin_amdgcn_wave_barrier();
        hid_dl_raw(wimg, lane, hb0, hb1);
        HID_LGKM0();
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<u32x4*>(wimg + hid_off16(m, 2 * kg)) = gv0;
        *reinterpret_cast<u32x4*>(wimg + hid_off16(m, 2 * kg + 1)) = gv1;
        HID_LGKM0();
        __builtin_amdgcn_wave_barrier();
        f32x2 G2[8];  // (pairs, not one 16-wide vector: element-wise updates of an f32x16 cost the compiler ~100 registers of copies)
        {
            u32x4 r0, r1;
            hid_dl_raw(wimg, lane, r0, r1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                G2[i] = f32x2{mtl_lo2<T>(r0[i]), mtl_hi2<T>(r0[i])};
                G2[4 + i] = f32x2{mtl_lo2<T>(r1[i]), mtl_hi2<T>(r1[i])};
            }
        }
        HID_LGKM0();
        __builtin_amdgcn_wave_barrier();
        f32x16 accQ;
#pragma unroll
        for (int e = 0; e < 16; ++e) accQ[e] = 0.f;
        auto task = [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            const u32x2 tbh = *reinterpret_cast<const u32x2*>(tabH + ((size_t)t * HC + cw + c) * 8);
            const u32x2 tbu = *reinterpret_cast<const u32x2*>(tabU + ((size_t)t * HC + cw + c) * 8);
            __builtin_amdgcn_sched_barrier(0);  // (tasks interleaved by the scheduler would need all their temporaries at once)
            f32x16 h, u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[2 * i] = mtl_lo2<T>(hb0[i]);
                h[2 * i + 1] = mtl_hi2<T>(hb0[i]);
                h[8 + 2 * i] = mtl_lo2<T>(hb1[i]);
                h[8 + 2 * i + 1] = mtl_hi2<T>(hb1[i]);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) u[e] = 0.f;
            {   // (the row block's P1 / Q2 values as A operands: re-read per task, 16 bytes each, instead of 8 registers held across the block)
                const u32x4 fp = *reinterpret_cast<const u32x4*>(pimg + hid_off16(m, kg));
                const u32x4 fq = *reinterpret_cast<const u32x4*>(pimg + hid_off16(m, 2 + kg));
                sp_mma1<T>(fp, hid_place<t>(tbh, kg), h);
                sp_mma1<T>(fq, hid_place<t>(tbu, kg), u);
            }
            u32x4 d0, d1, x0, x1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if ((i & 1) == 0) __builtin_amdgcn_sched_barrier(0);  // two pairs in flight: all eight at once cost ~100 registers of temporaries
                f32x2 av, gd;
                hid_gelu2(f32x2{h[2 * i], h[2 * i + 1]}, av, gd);
                const f32x2 dh = f32x2{u[2 * i], u[2 * i + 1]} * gd;
                G2[i] += dh;
                const uint32_t wd = hid_pk<T>(dh.x, dh.y), wa = hid_pk<T>(av.x, av.y);
                if (i < 4) {
                    d0[i] = wd;
                    x0[i] = wa;
                } else {
                    d1[i - 4] = wd;
                    x1[i - 4] = wa;
                }
            }
            // row reductions: A operands from the registers, B operands = the P1 / Q2 images in the same row order, task t's columns only
            // (ONE accumulator for both: columns n < 16 collect dB1^T, columns 16 + n dA2 -- the image holds P1 and Q2 side by side and
            // each product sees only task t's columns of its half)
            // d and x of a lane never meet: lanes n < 16 only need the dH operand's product, lanes n >= 16 the activation's.  Since the
            // B operand of lane n is zero outside (its half, task t), BOTH products can use the same masked fragment:
            //   acc[c][n] += sum_m dH[m][c] PQ[m][n] (n in P half)   and   acc[c][n] += sum_m a[m][c] PQ[m][n] (n in Q half)
            // are two MFMAs per k half whose B operands are the P-masked and the Q-masked fragment
            const uint32_t tmP = (n >> 2) == t ? 0xFFFFFFFFu : 0u, tmQ = (n >> 2) == t + 4 ? 0xFFFFFFFFu : 0u;
            __builtin_amdgcn_sched_barrier(0);
            {
                u32x4 pq0, pq1;
                hid_dl_raw(pimg, lane, pq0, pq1);
                u32x4 bP, bQ;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bP[i] = pq0[i] & tmP;
                    bQ[i] = pq0[i] & tmQ;
                }
                sp_mma1<T>(d0, bP, accF);
                sp_mma1<T>(x0, bQ, accF);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bP[i] = pq1[i] & tmP;
                    bQ[i] = pq1[i] & tmQ;
                }
                sp_mma1<T>(d1, bP, accF);
                sp_mma1<T>(x1, bQ, accF);
            }
            __builtin_amdgcn_sched_barrier(0);
            // column reduction Q1 += dH_t B1_t^T: rows along lanes through the wave's image
            hid_dl_store_t(wimg, lane, d0, d1);
            HID_LGKM0();
            __builtin_amdgcn_wave_barrier();
            u32x4 f0, f1;
            hid_tr_frag(wimg, lane, f0, f1);
            const int br = ((n >> 2) == t ? n : 16) * bs;
            const u32x4 fb0 = *reinterpret_cast<const u32x4*>(bq + br + (cw + 8 * hh) * 2);
            const u32x4 fb1 = *reinterpret_cast<const u32x4*>(bq + br + (cw + 16 + 8 * hh) * 2);
            sp_mma1<T>(f0, fb0, accQ);
            sp_mma1<T>(f1, fb1, accQ);
            __builtin_amdgcn_wave_barrier();
        };
        task(std::integral_constant<int, 0>{});
        task(std::integral_constant<int, 1>{});
        task(std::integral_constant<int, 2>{});
        task(std::integral_constant<int, 3>{});
        {   // G leaves as it came: transposed through the image, 16-byte row-major stores
            u32x4 g0, g1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g0[i] = hid_pk<T>(G2[i].x, G2[i].y);
                g1[i] = hid_pk<T>(G2[4 + i].x, G2[4 + i].y);
            }
            hid_dl_store_t(wimg, lane, g0, g1);
            HID_LGKM0();
            __builtin_amdgcn_wave_barrier();
            u32x4 f0, f1;
            hid_tr_frag(wimg, lane, f0, f1);  // lane (m, hh): columns 8 hh .. + 7 and 16 + 8 hh .. + 7 of row m
            if (live) {
                *reinterpret_cast<u32x4*>(gout + row * H + c_base + cw + 8 * hh) = f0;
                *reinterpret_cast<u32x4*>(gout + row * H + c_base + cw + 16 + 8 * hh) = f1;
            }
        }
        __syncthreads();
        if (n < 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((size_t)wave * 16 + n) * 33 + (r & 3) + 8 * (r >> 2) + 4 * hh] = accQ[r];
        }
        __syncthreads();
        for (int f = tid; f < 512; f += NW * 64) {
            const int rr = f & 31, nn = f >> 5;
            if (m0 + rr < P.M) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) a += red[((size_t)w * 16 + nn) * 33 + rr];
                P.rowpart[((int64_t)chunk * P.M + m0 + rr) * 16 + nn] = a;
            }
        }
        // (the next block's first barrier orders these reads before the images are written again)
    }
    // the wave's share of the factor gradients: lane (n, hh) holds columns cw + a(r) of rank column n
    if ((n & 15) < nt * 4) {
        const int t = (n & 15) >> 2, rho = n & 3, kind = n >> 4;
        float* part = P.part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * nt * 2 * 4 * HC;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = cw + (r & 3) + 8 * (r >> 2) + 4 * hh;
            part[((int64_t)(t * 2 + kind) * 4 + rho) * HC + j] = accF[r];
        }
    }
}

// (chunk, row) partial row sums -> alpha-scaled 16-byte rank segments of P2 / Q1 (columns past the rank written as zeros)
template <typename T>
__global__ __launch_bounds__(256) void k_hid_rows_finish(const float* __restrict__ rowpart, int n_chunk, int64_t M, int nt, const float* alpha,
                                                         HidOff off, T* out, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nt) return;
    const int64_t row = i / nt;
    const int t = (int)(i - row * nt);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < n_chunk; ++ch) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rowpart + ((int64_t)ch * M + row) * 16 + 4 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
    int o = 0;
#pragma unroll
    for (int q = 0; q < HID_TG; ++q)
        if (q == t) o = off.v[q];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] *= alpha[o + e];
    const u32x4 w = {mtl_pack2<T>(s[0], s[1]), mtl_pack2<T>(s[2], s[3]), 0u, 0u};
    *reinterpret_cast<u32x4*>(out + row * ldo + o) = w;
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
    // element i = (t, kind, rho, j) of the H-wide result; its partials live in the workgroups of chunk j / chunk_cols
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
    const int ch = j / P.chunk_cols, jl = j - ch * P.chunk_cols;
    const int64_t per_wg = (int64_t)P.nt * 2 * P.RR * P.chunk_cols;
    const float* src = P.part + (int64_t)ch * P.n_wg * per_wg + ((int64_t)(t * 2 + kind) * P.RR + rho) * P.chunk_cols + jl;
    float s0 = 0.f, s1 = 0.f;
    int w = 0;
    for (; w + 1 < P.n_wg; w += 2) {
        s0 += src[(int64_t)w * per_wg];
        s1 += src[(int64_t)(w + 1) * per_wg];
    }
    if (w < P.n_wg) s0 += src[(int64_t)w * per_wg];
    const float sv = s0 + s1;
    if (kind)
        dst[(int64_t)rho * P.H + j] = sv;
    else
        dst[(int64_t)j * r_t + rho] = sv;
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
        case 3: {
            const dim3 gd((unsigned)L.n_wg, (unsigned)L.n_chunk);
#define HID_D(K, HC_)                                                      \
    do {                                                                   \
        HID_RAISE_LDS((K<T, HC_>));                                        \
        hipLaunchKernelGGL((K<T, HC_>), gd, dim3(HC_ * 2), L.lds, s, q);   \
    } while (0)
            if (L.kind == 2) {
                if (L.hc == 384) HID_D(k_hid_fwd_d, 384);
                else if (L.hc == 128) HID_D(k_hid_fwd_d, 128);
                else HID_D(k_hid_fwd_d, 256);
            } else {
                if (L.hc == 384) HID_D(k_hid_bwd_d, 384);
                else if (L.hc == 128) HID_D(k_hid_bwd_d, 128);
                else HID_D(k_hid_bwd_d, 256);
            }
#undef HID_D
            break;
        }
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
MTL_INTERNAL void mtli_hid_rows_finish(int dtype, const float* rowpart, int n_chunk, int64_t M, int nt, const float* alpha, const int* off,
                                       void* out, int ldo, void* stream) {
    HidOff o;
    for (int i = 0; i < HID_TG; ++i) o.v[i] = off[i];
    const unsigned blocks = (unsigned)((M * nt + 255) / 256);
    if (dtype == MTLORA_F16)
        hipLaunchKernelGGL(k_hid_rows_finish<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowpart, n_chunk, M, nt, alpha, o,
                           reinterpret_cast<f16*>(out), ldo);
    else
        hipLaunchKernelGGL(k_hid_rows_finish<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowpart, n_chunk, M, nt, alpha, o,
                           reinterpret_cast<bf16*>(out), ldo);
}
MTL_INTERNAL void mtli_hid_reduce(const HidRedParams* r, int64_t per, void* stream) {
    hipLaunchKernelGGL(k_hid_reduce, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *r);
}
