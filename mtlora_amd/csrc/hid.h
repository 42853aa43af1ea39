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
#pragma once

constexpr int HID_TG = 4;   // tasks per launch (rank <= 4); launches with rank <= 8 take HID_TG / 2
constexpr int HID_RB = 4;   // rows between two workgroup barriers
constexpr int HID_MAXW = 16;

struct HidParams {
    const void* hbase;    // (M x H)
    const void* p1;       // (M x ldp1)  P of fc1; task i of this launch owns columns [off1[i], off1[i] + 8)
    void* p2;             // forward out (M x ldp2): columns [off2[i], off2[i] + 8) (alpha-scaled; columns past the rank are written as zeros)
    const void* q2;       // backward in (M x ldq2): Q of fc2, columns off2[i]
    void* q1;             // backward out (M x ldq1): Q of fc1, columns off1[i]
    const void* gsrc;     // backward in (M x H): dH_s for the first task group, the running G for the following ones (may be null: zeros)
    void* g;              // backward out (M x H)
    const void* b1t;      // (R1 x H) bt_cat of fc1: row rr = B1[:, rr], unscaled
    const void* a2;       // (R2 x H) a_cat of fc2: row rr = A2[rr, :], unscaled
    const float* alpha1;  // (R1)
    const float* alpha2;  // (R2)
    float* part;          // backward: [gridDim.x][nt][2][RR][H]  (kind 0: dB1^T[rho][j], kind 1: dA2[rho][j])
    int64_t M;
    int H, nt;
    int ldp1, ldp2, ldq1, ldq2;
    int off1[HID_TG], off2[HID_TG];
};

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
            b1[t][rho] = a2[t][rho] = f32x2{0.f, 0.f};
            if (t < nt) {
                const uint32_t wb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.b1t) + (int64_t)(P.off1[t] + rho) * H + c2);
                const uint32_t wa = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.a2) + (int64_t)(P.off2[t] + rho) * H + c2);
                b1[t][rho] = f32x2{mtl_lo2<T>(wb), mtl_hi2<T>(wb)};
                a2[t][rho] = f32x2{mtl_lo2<T>(wa), mtl_hi2<T>(wa)};
            }
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
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) v[t * RR + rho] = 0.f;
                if (t < nt) {
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
            }
            int idx;
            const float s = hid_wave_reduce<NV>(v, lane, idx);
            if (lane < NV) rd[(r * HID_MAXW + wave) * 16 + idx] = s;
        }
        __syncthreads();
        hid_finish_rows<T, TG, RR>(rd, NW, tid, rb * HID_RB, P.M, nt, P.alpha2, P.off2, reinterpret_cast<T*>(P.p2), P.ldp2);
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
            b1[t][rho] = a2[t][rho] = accB[t][rho] = accA[t][rho] = f32x2{0.f, 0.f};
            if (t < nt) {
                const uint32_t wb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.b1t) + (int64_t)(P.off1[t] + rho) * H + c2);
                const uint32_t wa = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(P.a2) + (int64_t)(P.off2[t] + rho) * H + c2);
                b1[t][rho] = f32x2{mtl_lo2<T>(wb), mtl_hi2<T>(wb)};
                a2[t][rho] = f32x2{mtl_lo2<T>(wa), mtl_hi2<T>(wa)};
            }
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
#pragma unroll
                for (int rho = 0; rho < RR; ++rho) v[t * RR + rho] = 0.f;
                if (t < nt) {
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

struct HidRedParams {
    const float* part;  // [n_wg][nt][2][RR][H]
    int n_wg, nt, RR, H;
    int r[HID_TG];         // un-padded ranks
    float* dB1[HID_TG];    // (H x r) row-major, nullable
    float* dA2[HID_TG];    // (r x H) row-major, nullable
};
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
