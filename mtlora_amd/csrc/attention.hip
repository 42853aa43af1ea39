// attention.hip -- (shifted-)window attention core, forward and backward, on CDNA4 (gfx950).
// Replaces swin_transformer_mtlora.py:194-220 and, with image_layout = 1, also the roll/partition
// (:336-350) before it and the merge/roll (:365-377) after it (folded into the addressing).
//
// One wave (one 64-thread workgroup) processes one (window, head) pair at a time:
//   N = ws*ws <= 64 tokens padded to 64, head_dim = 32  ->  every product is a handful of
//   32x32 MFMA tiles; nothing but Q/K/V(/dO) rows ever leaves registers/LDS (no (B_,nH,N,N)
//   score tensor in HBM -- the reference materialises it 3+ times).
//
// forward:   S^T[j][i] = K Q^T (lane owns query column i, registers span keys j)
//            softmax over keys = in-lane reduction + one exchange with lane^32
//            O^T[d][i]  = V^T P^T : V^T operand via ds_read_b64_tr_b16 on the row-major LDS image,
//                                   P^T operand straight from the accumulator registers.
// backward:  pass 1 (query-owned):  P^T, dP^T = V dO^T, D_i, dS^T -> dbias (REGISTER accumulate: the (key, query) a
//                                   lane's accumulator element maps to is the same for every window), dQ^T = K^T dS^T
//            pass 2 (key-owned):    S, P, dP recomputed in the other orientation from the row stats of
//                                   pass 1 -> dK^T = Q^T dS, dV^T = dO^T P
// Both kernels are persistent (grid = resident workgroups) and software-pipelined: the NEXT window's Q/K/V(/dO) rows
// are loaded into registers while the current window is multiplied; token addresses use a per-window decomposition
// plus per-lane constants (no per-token division); the per-element sections are branch-free (clamped indices +
// selects) so that their LDS reads are issued back to back; the dense-mask path is a separate template variant.
// The k-slot <-> key assignment of every register-fed operand follows the accumulator layout
// (row = (r&3) + 8(r>>2) + 4(lane>>5)); the LDS-fed partner operand is gathered in the same order.
#include "common.h"

namespace {

#ifndef MTL_ATTN_FWD_WPS
#define MTL_ATTN_FWD_WPS 2  // waves per SIMD the forward kernel is compiled for
#endif
#ifndef MTL_ATTN_BWD_WPS
#define MTL_ATTN_BWD_WPS 1  // waves per SIMD the backward kernel is compiled for (register budget 512 / WPS; at 2 it spills 254 VGPRs)
#endif
constexpr int AN = 64;  // padded tokens per window
constexpr int HD = 32;  // head_dim

template <typename T>
struct AC;
template <>
struct AC<bf16> {
    static constexpr int KT_D = 1;  // k-tiles across head_dim (32 bf16 = 64 B)
    static constexpr int KT_N = 2;  // k-tiles across the 64 padded tokens
    static constexpr int RS = 80;   // LDS row stride in bytes (64 + 16)
    static constexpr int VPR = 4;   // 16-byte vectors per row
};
template <>
struct AC<f16> : AC<bf16> {};  // same 16-bit geometry
template <>
struct AC<float> {
    static constexpr int KT_D = 2;
    static constexpr int KT_N = 4;
    static constexpr int RS = 144;  // 128 + 16
    static constexpr int VPR = 8;
};

struct AttnParams {
    const void* qkv;
    const float* bias;     // (nH, N, N) [h][i][j]
    const float* mask;     // (nWimg, N, N) [w][i][j] or null  (general additive mask, slow path)
    const int* mask_ids;   // (nWimg, N) region id per token or null: mask[i][j] = ids differ ? mask_value : 0
    float mask_value;
    void* out;
    const void* dout;
    void* dqkv;
    float* dbias_part;  // [G][nH][N(j)][N(i)]
    int64_t n_windows;
    int H, W, ws, shift, nH, N, C, nWx, nWy, image_layout;
    int G;
    float scale;
};

// Token addressing without per-token divisions.  A window id is decomposed ONCE per window (wave-uniform, 32-bit);
// a token's (ty, tx) inside the window is a per-lane constant computed before the persistent loop; the cyclic shift
// wraps with one conditional subtract (wy*ws + ty + shift < H + ws).  [64-bit div / mod per token per window was
// ~800 instructions a call.]
struct WinPos {
    int64_t base;  // windows layout: first token of the window; image layout: b * H * W
    int y0, x0;    // image layout: wy*ws + shift, wx*ws + shift
};
__device__ __forceinline__ WinPos win_pos(const AttnParams& p, int64_t w) {
    WinPos q;
    if (!p.image_layout) {
        q.base = w * p.N;
        q.y0 = q.x0 = 0;
        return q;
    }
    const uint32_t wu = (uint32_t)w;  // n_windows < 2^31 (checked by the host)
    const uint32_t wx = wu % (uint32_t)p.nWx, r = wu / (uint32_t)p.nWx;
    const uint32_t wy = r % (uint32_t)p.nWy, b = r / (uint32_t)p.nWy;
    q.base = (int64_t)b * p.H * p.W;
    q.y0 = (int)wy * p.ws + p.shift;
    q.x0 = (int)wx * p.ws + p.shift;
    return q;
}
// t = ty * ws + tx  (tyx packs ty << 8 | tx; t itself is kept for the windows layout)
__device__ __forceinline__ int pack_tyx(const AttnParams& p, int t) {
    const int ty = t / p.ws;
    return (ty << 8) | (t - ty * p.ws);
}
// index of token t of the window RELATIVE to q.base (the window's first token / the image's first token): 32 bits, so that every
// per-lane address is [wave-uniform 64-bit base] + [32-bit offset] (the host checks H * W * 3C * elem_size < 2^31) -- 64-bit
// multiplies per row and output vector were a tenth of the kernels' VALU instructions
__device__ __forceinline__ int token_at(const AttnParams& p, const WinPos& q, int t, int tyx) {
    if (!p.image_layout) return t;
    int y = q.y0 + (tyx >> 8), x = q.x0 + (tyx & 255);
    y = y >= p.H ? y - p.H : y;
    x = x >= p.W ? x - p.W : x;
    return y * p.W + x;
}

// The LDS image of a (window, head) operand holds its N token rows plus ONE shared zero row (index N): every
// padded row 64 > r >= N of the 64-row MFMA tiles reads that row (clamped index), so an image costs
// (N + 1) * RS bytes instead of 64 * RS.
__device__ __forceinline__ int img_rows(int N) { return N < AN ? N + 1 : AN; }
__host__ __device__ __forceinline__ int img_rows_h(int N) { return N < AN ? N + 1 : AN; }

// which (row, 16-byte vector) of an image lane `lane` copies in iteration `it`.  ds_write_b128 is served 8 lanes (128 bytes =
// all 32 banks) per LDS cycle, so 8 consecutive lanes must hit 8 distinct 16-byte pieces mod 128 B.  bf16 rows hold 4 vectors
// at an 80-byte stride (5 pieces: odd, which keeps the ds_read_b128 fragment reads conflict-free): rows r and r + 1 collide
// on one piece (2-way conflict on every staging store: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.32 in k_attn_bwd, round 1),
// rows r and r + 4 (20 pieces apart = 4 mod 8) interleave exactly.  f32 rows are 8 vectors: one row per 8 lanes already.
template <typename T>
__device__ __forceinline__ void av_map(int it, int lane, int& row, int& vec) {
    constexpr int VPR = AC<T>::VPR;
    if constexpr (VPR == 4) {
        const int g = lane >> 3, l = lane & 7;
        row = it * 16 + (g >> 2) * 8 + (g & 3) + 4 * (l >> 2);
        vec = l & 3;
    } else {
        const int idx = it * 64 + lane;
        row = idx / VPR;
        vec = idx % VPR;
    }
}

// the same copy split in two, for the software pipeline of the persistent kernels: the next window's rows are loaded
// into registers while the current window is being multiplied, and written to the images at the next iteration
template <typename T>
struct RowRegs {
    u32x4 v[AC<T>::VPR];
};
template <typename T>
struct RowIds {
    int t[AC<T>::VPR], tyx[AC<T>::VPR];  // token (row) this lane copies in iteration it, -1 = none
};
template <typename T>
__device__ __forceinline__ RowIds<T> row_ids(const AttnParams& p, int lane) {
    RowIds<T> r;
#pragma unroll
    for (int it = 0; it < AC<T>::VPR; ++it) {
        int row, vec;
        av_map<T>(it, lane, row, vec);
        (void)vec;
        r.t[it] = row < p.N ? row : -1;
        r.tyx[it] = row < p.N ? pack_tyx(p, row) : 0;
    }
    return r;
}
// off[it]: ELEMENT offset (token index * stride + the lane's vector) of the row this lane copies in iteration it, relative to the
// window's base pointer; -1 = none
template <typename T>
__device__ __forceinline__ void row_offsets(int (&off)[AC<T>::VPR], const AttnParams& p, const WinPos& q, const RowIds<T>& ids,
                                            int stride, int lane) {
#pragma unroll
    for (int it = 0; it < AC<T>::VPR; ++it) {
        int row, vec;
        av_map<T>(it, lane, row, vec);
        (void)row;
        off[it] = ids.t[it] >= 0 ? token_at(p, q, ids.t[it], ids.tyx[it]) * stride + vec * ET<T>::VEC : -1;
    }
}
// base: wave-uniform pointer (tensor + window base + head / q-k-v slice)
template <typename T>
__device__ __forceinline__ void load_rows(RowRegs<T>& r, const T* base, const int (&off)[AC<T>::VPR]) {
#pragma unroll
    for (int it = 0; it < AC<T>::VPR; ++it)
        r.v[it] = off[it] >= 0 ? *reinterpret_cast<const u32x4*>(base + (uint32_t)off[it]) : u32x4{0u, 0u, 0u, 0u};
}
template <typename T>
__device__ __forceinline__ void store_rows(unsigned char* s, const RowRegs<T>& r, int n, int lane) {
    constexpr int VPR = AC<T>::VPR, RS = AC<T>::RS;
    const int rows = img_rows(n);
#pragma unroll
    for (int it = 0; it < VPR; ++it) {
        int row, vec;
        av_map<T>(it, lane, row, vec);
        if (row < rows) *reinterpret_cast<u32x4*>(s + row * RS + vec * 16) = r.v[it];
    }
}

// operand whose MFMA rows are the LDS image rows sub*32 + (lane&31), k = columns of k-tile kt
template <typename T>
__device__ __forceinline__ Frag<T> rowfrag(const unsigned char* s, int sub, int kt, int lane, int n) {
    const int h = lane >> 5;
    int row = sub * 32 + (lane & 31);
    row = row < n ? row : n;  // padded rows -> the shared zero row
    const unsigned char* p = s + row * AC<T>::RS + kt * 64;
    Frag<T> f;
    f.v[0] = *reinterpret_cast<const u32x4*>(p + h * 16);
    f.v[1] = *reinterpret_cast<const u32x4*>(p + (2 + h) * 16);
    return f;
}

// operand whose MFMA rows are the 32 COLUMNS (d) of the image, k = image rows of k-tile kt, slot order =
// accumulator order: slot (h, r) <-> row kt*KE + (r&3) + 8(r>>2) + 4h
template <typename H>  // 16-bit element types: the transposing read moves bits
__device__ __forceinline__ Frag<H> colfrag16(const unsigned char* s, int kt, int lane, int n) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int col = 16 * (g & 1) + 4 * (i & 3);
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = kt * 32 + 8 * q + 4 * h + (i >> 2);
        row = row < n ? row : n;
        const unsigned char* p = s + row * AC<bf16>::RS + col * 2;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        u32x2 u = __builtin_bit_cast(u32x2, v);
        w[2 * q] = u[0];
        w[2 * q + 1] = u[1];
    }
    Frag<H> f;
    f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
    f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
    return f;
}
__device__ __forceinline__ Frag<bf16> colfrag(const unsigned char* s, int kt, int lane, int n, bf16*) {
    return colfrag16<bf16>(s, kt, lane, n);
}
__device__ __forceinline__ Frag<f16> colfrag(const unsigned char* s, int kt, int lane, int n, f16*) {
    return colfrag16<f16>(s, kt, lane, n);
}
__device__ __forceinline__ Frag<float> colfrag(const unsigned char* s, int kt, int lane, int n, float*) {
    const int h = lane >> 5, d = lane & 31;
    uint32_t w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int row = kt * 16 + (e & 3) + 8 * (e >> 2) + 4 * h;
        row = row < n ? row : n;
        w[e] = *reinterpret_cast<const uint32_t*>(s + row * AC<float>::RS + d * 4);
    }
    Frag<float> f;
    f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
    f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
    return f;
}

constexpr float NEG_BIG = -1.0e30f;

// P / dS hand-over images of the backward ([32 query rows][64 keys], row stride IMG_RS): the accumulators hold them
// query-per-lane; dK / dV need them key-per-lane with the queries along k, i.e. transposed -- written row-wise, read back
// with the transposing load exactly like colfrag (same slot <-> row order as the Q / dO operand they are paired with)
template <typename T>
struct IMG;
template <>
struct IMG<bf16> {
    // 152 B = 38 dwords: the 32 query rows of an img_store (8 bytes per lane) start at 38 i mod 64 = every even bank exactly once
    // -> the 64 banks are hit once per wave-instruction.  With 144 B (36 dwords) rows i and i + 16 shared their banks: a 2-way
    // conflict on every P / dS hand-over store (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.25 for k_attn_bwd,
    // profiles/r02_pmc_sq_after_attn_staging_map.csv).  The transposing reads (4 rows x 32 B per 16-lane group) stay disjoint.
    static constexpr int RS = 64 * 2 + 24;
};
template <>
struct IMG<f16> : IMG<bf16> {};
template <>
struct IMG<float> {
    static constexpr int RS = 64 * 4 + 16;
};
// lane (i = lane & 31, h = lane >> 5) writes query row i: for each key block sj its 4 runs of 4 consecutive keys
template <typename T>
__device__ __forceinline__ void img_store(unsigned char* img, const f32x16 (&a)[2], int lane) {
    const int il = lane & 31, h = lane >> 5;
#pragma unroll
    for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j0 = sj * 32 + 8 * q + 4 * h;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(img + il * IMG<float>::RS + j0 * 4) =
                    f32x4{a[sj][4 * q], a[sj][4 * q + 1], a[sj][4 * q + 2], a[sj][4 * q + 3]};
            } else {
                *reinterpret_cast<u32x2*>(img + il * IMG<bf16>::RS + j0 * 2) =
                    u32x2{mtl_pk2<T>(a[sj][4 * q], a[sj][4 * q + 1]), mtl_pk2<T>(a[sj][4 * q + 2], a[sj][4 * q + 3])};
            }
        }
}
// MFMA rows = image columns [col0, col0 + 32), k = image rows of k-tile kt
template <typename H>
__device__ __forceinline__ Frag<H> imgfrag16(const unsigned char* s, int col0, int kt, int lane) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int col = col0 + 16 * (g & 1) + 4 * (i & 3);
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = kt * 32 + 8 * q + 4 * h + (i >> 2);
        const unsigned char* p = s + row * IMG<bf16>::RS + col * 2;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        u32x2 u = __builtin_bit_cast(u32x2, v);
        w[2 * q] = u[0];
        w[2 * q + 1] = u[1];
    }
    Frag<H> f;
    f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
    f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
    return f;
}
__device__ __forceinline__ Frag<bf16> imgfrag(const unsigned char* s, int col0, int kt, int lane, bf16*) {
    return imgfrag16<bf16>(s, col0, kt, lane);
}
__device__ __forceinline__ Frag<f16> imgfrag(const unsigned char* s, int col0, int kt, int lane, f16*) {
    return imgfrag16<f16>(s, col0, kt, lane);
}
__device__ __forceinline__ Frag<float> imgfrag(const unsigned char* s, int col0, int kt, int lane, float*) {
    const int h = lane >> 5, d = lane & 31;
    uint32_t w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int row = kt * 16 + (e & 3) + 8 * (e >> 2) + 4 * h;
        w[e] = *reinterpret_cast<const uint32_t*>(s + row * IMG<float>::RS + (col0 + d) * 4);
    }
    Frag<float> f;
    f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
    f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
    return f;
}

// operand fed from accumulator registers: acc[0..1] are the two 32-row subtiles along the k axis
__device__ __forceinline__ Frag<bf16> regfrag(const f32x16 (&acc)[2], int kt, bf16*) {
    Frag<bf16> f;
    const f32x16& a = acc[kt];
    f.v[0] = u32x4{mtl_pk2<bf16>(a[0], a[1]), mtl_pk2<bf16>(a[2], a[3]), mtl_pk2<bf16>(a[4], a[5]), mtl_pk2<bf16>(a[6], a[7])};
    f.v[1] = u32x4{mtl_pk2<bf16>(a[8], a[9]), mtl_pk2<bf16>(a[10], a[11]), mtl_pk2<bf16>(a[12], a[13]), mtl_pk2<bf16>(a[14], a[15])};
    return f;
}
__device__ __forceinline__ Frag<f16> regfrag(const f32x16 (&acc)[2], int kt, f16*) {
    Frag<f16> f;
    const f32x16& a = acc[kt];
    f.v[0] = u32x4{mtl_pk2<f16>(a[0], a[1]), mtl_pk2<f16>(a[2], a[3]), mtl_pk2<f16>(a[4], a[5]), mtl_pk2<f16>(a[6], a[7])};
    f.v[1] = u32x4{mtl_pk2<f16>(a[8], a[9]), mtl_pk2<f16>(a[10], a[11]), mtl_pk2<f16>(a[12], a[13]), mtl_pk2<f16>(a[14], a[15])};
    return f;
}
__device__ __forceinline__ Frag<float> regfrag(const f32x16 (&acc)[2], int kt, float*) {
    Frag<float> f;
    const f32x16& a = acc[kt >> 1];
    const int o = (kt & 1) * 8;
    f.v[0] = __builtin_bit_cast(u32x4, f32x4{a[o + 0], a[o + 1], a[o + 2], a[o + 3]});
    f.v[1] = __builtin_bit_cast(u32x4, f32x4{a[o + 4], a[o + 5], a[o + 6], a[o + 7]});
    return f;
}

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// store a [d][token] accumulator subtile (lane owns token i = lane & 31, registers 4q..4q+3 are d = 8q + 4h + 0..3) through a per-wave LDS image: the accumulator subtile holds [d][token] with a lane owning ONE token and 4-element
// runs of d, so a direct store writes 8-byte pieces of 32 different rows per instruction (16-byte pieces per token with its h
// partner) -- partial-sector writes that the memory system serves at half the rate of whole 64-byte row segments (ablation
// tools/ab: the forward's loads + stores alone took 79 of its 92 us at stage 0).  Here the subtile is written to the image
// token-major, read back 16 bytes per lane (4 lanes = one token's 32-element head slice) and stored with whole row segments per
// 4 lanes.  `tokrow`: LDS table of the 32 tokens' element offsets; rows >= n_rows are padding.
template <typename T>
struct STG {
    static constexpr int RB = 32 * (int)sizeof(T);  // bytes of a token's head slice
    static constexpr int RS = RB + 16;              // image row stride
    static constexpr int BYTES = 32 * RS;
    static constexpr int PPR = RB / 16;             // 16-byte pieces per row
};
template <typename T>
__device__ __forceinline__ void store_dt_lds(unsigned char* stg, T* base, const int* tokrow, int n_rows, const f32x16& a, float mul,
                                             int lane) {
    const int il = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned char* dst = stg + il * STG<T>::RS + (8 * q + 4 * h) * (int)sizeof(T);
        if constexpr (sizeof(T) == 4)
            *reinterpret_cast<f32x4*>(dst) = f32x4{a[4 * q] * mul, a[4 * q + 1] * mul, a[4 * q + 2] * mul, a[4 * q + 3] * mul};
        else
            *reinterpret_cast<u32x2*>(dst) = u32x2{mtl_pk2<T>(a[4 * q] * mul, a[4 * q + 1] * mul), mtl_pk2<T>(a[4 * q + 2] * mul, a[4 * q + 3] * mul)};
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is private to this wave
    __builtin_amdgcn_wave_barrier();
    constexpr int PPR = STG<T>::PPR, IT = 32 * PPR / 64;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = it * 64 + lane, row = idx / PPR, pc = idx % PPR;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * STG<T>::RS + pc * 16);
        const int off = tokrow[row];
        if (row < n_rows) *reinterpret_cast<u32x4*>(base + (uint32_t)(off + pc * (16 / (int)sizeof(T)))) = v;
    }
    __builtin_amdgcn_wave_barrier();  // (the next user of the image writes it only after these reads have been issued in order)
}

// ---- the relative-position bias of the wave's head.  A wave serves ONE head for all of its windows, and the (key, query) pair behind
// accumulator element (sj, r) of lane l is the same for every window.
//  * 16-bit types: the bias enters through the MATRIX pipe.  scores' = K Q^T + I . (bias / scale)^T -- an identity A operand times a B
//    operand holding bias / scale in the compute type (32 VGPRs per wave for the whole launch, loaded once): 8 extra MFMAs per window
//    and head on a pipe that is 90 % idle here, and the per-element section of round 1-3 (an LDS read of the bias image, a select for the
//    padded keys, an FMA: ~450 VALU + 64 LDS instructions per window and head in a VALU-issue-bound kernel) disappears; the softmax
//    runs on scores' with the multiplier scale * log2(e) folded into its exponent FMA.  Padded KEYS (j >= N) carry PAD_BIG in the bias
//    operand, padded QUERIES (never stored) 0.  Round 5: the bias product runs as an FP16 MFMA in the bf16 kernels too
//    (v_mfma_f32_32x32x16_f16 with an fp16 identity: the two operands of THIS product are independent of the q / k type) -- rounded to
//    bf16, a trained table (|bias| up to ~15) put 2^-9 |bias| / scale ~ 0.03 of absolute error on a logit and the forward missed the
//    1e-2 tolerance against the fp64 oracle (1.3e-2, `test_attention_large_relative_bias_vs_oracle`); fp16 carries 11 bits, and
//    bias / scale <= 15 * 5.7 is far inside its range.  (The reference adds the fp32 bias to a bf16-ROUNDED q k^T,
//    swin_transformer_mtlora.py:200-207.)
//    The 9.6 KB bias image per workgroup is gone from LDS as well.
//  * fp32: exact path as before -- an LDS image of bias[head] with an odd row stride, one read + FMA + select per element.
constexpr int bias_stride_c(int N) { return (N & 1) ? N : N + 1; }
__device__ __forceinline__ int bias_stride(int N) { return (N & 1) ? N : N + 1; }

template <typename T>
struct BiasSrc {  // 16-bit
    // (both operands of the bias product are FP16 whatever T is -- see the note above: 11 mantissa bits instead of bf16's 8)
    Frag<f16> idf;      // identity: lane (j = lane & 31, h), k-slot (t, e) <-> k = 16 t + 8 h + e:  1 where k == j
    Frag<f16> bf[2][2]; // [si][sj]: lane (i = lane & 31, h), k-slot (t, e):  bias[32 si + i][32 sj + 16 t + 8 h + e] / scale
    float c;            // softmax multiplier on scores': scale * log2(e)
    float maskv;        // shift-mask value in the scores' domain: mask_value / scale
};
template <>
struct BiasSrc<float> {
    const float* sBias;  // LDS image [N][bias_stride(N)]
    float c;             // log2(e)
    float maskv;
};

template <typename T>
__device__ __forceinline__ float pad_big() {  // value of a padded key in the (fp16) bias operand
    if constexpr (__is_same(T, f16))
        return -3.0e4f;   // (fp16 q / k: times scale * log2 e it is still far below any real score)
    else
        return -6.0e4f;   // bf16 q / k: the fp16 operand's most negative useful value (a scaled logit of -10 600)
}

template <typename T>
__device__ __forceinline__ void bias_init(BiasSrc<T>& B, const AttnParams& p, int head, float* sBias, int lane) {
    const float* src = p.bias + (int64_t)head * p.N * p.N;
    if constexpr (sizeof(T) == 4) {
        const int bs = bias_stride(p.N);
        for (int idx = lane; idx < p.N * p.N; idx += 64) sBias[(idx / p.N) * bs + (idx % p.N)] = src[idx];
        B.sBias = sBias;
        B.c = 1.4426950408889634f;
        B.maskv = p.mask_value;
    } else {
        (void)sBias;
        const float inv = 1.f / p.scale;
        const int rl = lane & 31, h = lane >> 5;
        const uint32_t one = 0x3C00u;  // fp16 1.0
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint32_t w[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const int k0 = 16 * t + 8 * h + 2 * e2;
                w[e2] = (k0 == rl ? one : 0u) | (k0 + 1 == rl ? one << 16 : 0u);
            }
            B.idf.v[t] = u32x4{w[0], w[1], w[2], w[3]};
        }
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            const int i = si * 32 + rl;
#pragma unroll
            for (int sj = 0; sj < 2; ++sj)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int j = sj * 32 + 16 * t + 8 * h + e;
                        f[e] = j < p.N ? (i < p.N ? src[i * p.N + j] * inv : 0.f) : pad_big<T>();
                    }
                    B.bf[si][sj].v[t] = u32x4{mtl_pk2<f16>(f[0], f[1]), mtl_pk2<f16>(f[2], f[3]), mtl_pk2<f16>(f[4], f[5]), mtl_pk2<f16>(f[6], f[7])};
                }
        }
        B.c = p.scale * 1.4426950408889634f;
        B.maskv = p.mask_value * inv;
    }
}

// scores for query half `si` in the query-owned orientation: st[sj][r] = S^T[key][query] (16-bit types: in the scores' = S / scale
// domain, see BiasSrc).  `masked` is wave-uniform: only the windows that straddle the shift boundary carry a mask (the last window row /
// column of an image: 31 of 256 at stage 0, and no window of an unshifted block) -- all the others skip the per-element mask section.
template <typename T, bool DENSE>
__device__ __forceinline__ void scores_t(f32x16 (&st)[2], const unsigned char* sK, const unsigned char* sQ, int si,
                                         const AttnParams& p, const BiasSrc<T>& B, const int* srid, bool masked, int wm, int lane) {
    zero(st[0]);
    zero(st[1]);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) mtl_mma(B.idf, B.bf[si][sj], st[sj]);
    }
#pragma unroll
    for (int kt = 0; kt < AC<T>::KT_D; ++kt) {
        Frag<T> fq = rowfrag<T>(sQ, si, kt, lane, p.N);
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
            Frag<T> fk = rowfrag<T>(sK, sj, kt, lane, p.N);
            mtl_mma(fk, fq, st[sj]);
        }
    }
    const int i = si * 32 + (lane & 31);
    const bool iv = i < p.N;
    const int ic = iv ? i : p.N - 1;
    const int h4 = 4 * (lane >> 5);
    if constexpr (sizeof(T) == 4) {
        // bias (+ mask), branch-free: clamped indices + selects, so the LDS reads of a lane are issued back to back
        const float* brow = B.sBias + ic * bias_stride(p.N);
        const int rid_i = srid[ic];  // 0 everywhere when there is no region-id mask
        const float* mrow = DENSE ? p.mask + ((int64_t)wm * p.N + ic) * p.N : nullptr;  // dense additive mask: slow path
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
            __builtin_amdgcn_sched_barrier(0);  // 16 elements' LDS reads in flight at a time, not 32 (register pressure)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = sj * 32 + (r & 3) + 8 * (r >> 2) + h4;
                const int jc = j < p.N ? j : p.N - 1;
                float add = brow[jc];
                if constexpr (DENSE)
                    add += mrow[jc];
                else
                    add += srid[jc] != rid_i ? B.maskv : 0.f;
                const float v = st[sj][r] * p.scale + (iv ? add : 0.f);
                st[sj][r] = j < p.N ? v : NEG_BIG;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        (void)masked;
    } else {
        if (DENSE || masked) {
            const int rid_i = srid[ic];
            const float* mrow = DENSE ? p.mask + ((int64_t)wm * p.N + ic) * p.N : nullptr;
            const float inv = DENSE ? 1.f / p.scale : 0.f;
#pragma unroll
            for (int sj = 0; sj < 2; ++sj) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = sj * 32 + (r & 3) + 8 * (r >> 2) + h4;
                    const int jc = j < p.N ? j : p.N - 1;
                    float add;
                    if constexpr (DENSE)
                        add = mrow[jc] * inv;
                    else
                        add = srid[jc] != rid_i ? B.maskv : 0.f;
                    st[sj][r] += (iv && j < p.N) ? add : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// in-place softmax numerators over the 64 keys of a query column (32 in this lane, 32 in lane^32): st <- exp2((st - max) * c),
// returns 1 / sum.  NORM: also multiply the numerators by it (the backward needs P itself; the 16-bit forward scales its 16 outputs
// instead).  EXACT (fp32 kernels): exp2((st - max) * c), the difference formed first -- the arithmetic of rounds 1-3 (__expf(st - max));
// the 16-bit kernels fold the subtraction into one FMA, st * c - max * c (relative error 2^-24 |st c| on the exponent: 1e-6).
template <bool NORM, bool EXACT>
__device__ __forceinline__ float softmax_t(f32x16 (&st)[2], float c) {
    float m = -3.0e38f;
#pragma unroll
    for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, st[sj][r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    const float mc = -m * c;
    float l = 0.f;
#pragma unroll
    for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = EXACT ? __builtin_amdgcn_exp2f((st[sj][r] - m) * c) : __builtin_amdgcn_exp2f(st[sj][r] * c + mc);
            st[sj][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if constexpr (NORM) {
#pragma unroll
        for (int sj = 0; sj < 2; ++sj)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[sj][r] *= inv;
    }
    return inv;
}

__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
    const int64_t q = n / 8, r = n % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

template <typename T, bool DENSE>
__global__ __launch_bounds__(64, MTL_ATTN_FWD_WPS) void k_attn_fwd(const AttnParams p) {
    constexpr int RS = AC<T>::RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int IB = ((img_rows(p.N) * RS + 15) / 16) * 16;  // bytes per image
    unsigned char* sQ = smem;
    unsigned char* sK = smem + IB;
    unsigned char* sV = smem + 2 * IB;
    int* tok = reinterpret_cast<int*>(smem + 3 * IB);
    int* srid = tok + AN;
    const int lane = threadIdx.x;
    const int64_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int head = (int)(L % p.nH);
    const int g = (int)(L / p.nH);
    const int nWimg = p.nWx * p.nWy;
    const T* qkv = reinterpret_cast<const T*>(p.qkv) + head * HD;
    T* out = reinterpret_cast<T*>(p.out) + head * HD;
    const int C3 = 3 * p.C;
    unsigned char* stg = reinterpret_cast<unsigned char*>(srid + AN);  // output staging image
    BiasSrc<T> B;
    bias_init<T>(B, p, head, reinterpret_cast<float*>(stg + STG<T>::BYTES), lane);
    const RowIds<T> ids = row_ids<T>(p, lane);
    const int lane_tyx = pack_tyx(p, lane < p.N ? lane : 0);
    RowRegs<T> rq, rk, rv;
    int rid_next = 0;  // region id of this lane's token in the prefetched window (a load at the top of the iteration, right
                       // in front of its LDS store, exposed one L2 round trip per window of every shifted block)
    auto prefetch = [&](int64_t w) __attribute__((always_inline)) {
        int off[AC<T>::VPR];
        const WinPos q = win_pos(p, w);
        row_offsets<T>(off, p, q, ids, C3, lane);
        const T* wb = qkv + q.base * C3;  // wave-uniform
        load_rows<T>(rq, wb, off);
        load_rows<T>(rk, wb + p.C, off);
        load_rows<T>(rv, wb + 2 * p.C, off);
        rid_next = (p.mask_ids && lane < p.N) ? p.mask_ids[(int)(w % nWimg) * p.N + lane] : 0;
    };
    if (g < p.n_windows) prefetch(g);

    for (int64_t w = g; w < p.n_windows; w += p.G) {
        const int wm = (int)(w % nWimg);
        const WinPos wq = win_pos(p, w);
        __syncthreads();  // previous item's LDS reads are done
        tok[lane] = lane < p.N ? token_at(p, wq, lane, lane_tyx) * p.C : 0;  // element offset of the token's output row
        srid[lane] = rid_next;
        // does this window carry a shift mask at all?  (wave-uniform: region ids of its tokens differ)
        const bool masked = __ballot(lane < p.N && rid_next != __builtin_amdgcn_readfirstlane(rid_next)) != 0ull;
        store_rows<T>(sQ, rq, p.N, lane);
        store_rows<T>(sK, rk, p.N, lane);
        store_rows<T>(sV, rv, p.N, lane);
        __syncthreads();
        if (w + p.G < p.n_windows) prefetch(w + p.G);  // next window's rows fly while this one is multiplied
        T* ob = out + wq.base * p.C;  // wave-uniform
        auto half = [&](int si) __attribute__((always_inline)) {
            f32x16 st[2];
            scores_t<T, DENSE>(st, sK, sQ, si, p, B, srid, masked, wm, lane);
            // 16-bit: numerators only, the 16 outputs are scaled by 1 / sum instead; fp32: P itself, as in rounds 1-3
            const float inv_raw = softmax_t<sizeof(T) == 4, sizeof(T) == 4>(st, B.c);
            const float inv_l = sizeof(T) == 4 ? 1.f : inv_raw;
            f32x16 o;
            zero(o);
#pragma unroll
            for (int kt = 0; kt < AC<T>::KT_N; ++kt) {
                Frag<T> fv = colfrag(sV, kt, lane, p.N, (T*)nullptr);
                Frag<T> fp = regfrag(st, kt, (T*)nullptr);
                mtl_mma(fv, fp, o);
            }
            store_dt_lds<T>(stg, ob, tok + si * 32, p.N - si * 32, o, inv_l, lane);
        };
        if constexpr (sizeof(T) == 2) {  // (the bias operand is indexed by the half: constant after inlining)
            half(0);
            if (p.N > 32) half(1);
        } else {
#pragma unroll 1
            for (int si = 0; si < 2; ++si) {
                if (si * 32 >= p.N) break;
                half(si);
            }
        }
    }
}

// One wave per workgroup and (LDS-limited) one workgroup per SIMD: the whole 512-entry register file is this wave's,
// so the next window's Q / K / V / dO rows are prefetched into registers while the current window is multiplied.
// Single pass over the two 32-query halves: S^T, P^T, dP^T, D, dS^T, dbias, dQ as before (query per lane), then P^T and
// dS^T of the half are handed over through two small LDS images and read back TRANSPOSED (key per lane, queries along k)
// to accumulate dK^T += Q^T dS and dV^T += dO^T P -- instead of a second, key-owned pass that recomputed S, P and dP from
// the row statistics (16 MFMAs, 64 exps per lane and a long element-wise section per window).
template <typename T, bool DENSE>
__global__ __launch_bounds__(64, MTL_ATTN_BWD_WPS) void k_attn_bwd(const AttnParams p) {
    constexpr int RS = AC<T>::RS;
    constexpr int KQ = 32 / (AC<T>::KT_N == 2 ? 32 : 16);  // k-tiles per 32-query half (bf16: 1, f32: 2)
    // Q, K, V, dO images + P / dS hand-over images + token table + region ids + bias[head]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int IB = ((img_rows(p.N) * RS + 15) / 16) * 16;  // bytes per image
    unsigned char* sQ = smem;
    unsigned char* sK = smem + IB;
    unsigned char* sV = smem + 2 * IB;
    unsigned char* sO = smem + 3 * IB;
    unsigned char* sPi = smem + 4 * IB;                 // [32][64] P of the current query half
    unsigned char* sDi = sPi + 32 * IMG<T>::RS;         // [32][64] dS
    int* tok = reinterpret_cast<int*>(sDi + 32 * IMG<T>::RS);
    int* srid = tok + AN;
    const int lane = threadIdx.x;
    const int64_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int head = (int)(L % p.nH);
    const int g = (int)(L / p.nH);
    const int nWimg = p.nWx * p.nWy;
    const T* qkv = reinterpret_cast<const T*>(p.qkv) + head * HD;
    const T* dout = reinterpret_cast<const T*>(p.dout) + head * HD;
    T* dqkv = reinterpret_cast<T*>(p.dqkv) + head * HD;
    const int C3 = 3 * p.C;
    unsigned char* stg = reinterpret_cast<unsigned char*>(srid + AN);  // output staging image
    BiasSrc<T> B;
    bias_init<T>(B, p, head, reinterpret_cast<float*>(stg + STG<T>::BYTES), lane);
    // dbias accumulator in REGISTERS: element (sj, r) of lane l is always (key j = 32 sj + row(l, r), query i = 32 si + l % 32),
    // the same pair for every window this wave visits
    f32x16 dbacc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) zero(dbacc[a][b]);
    const int h4 = 4 * (lane >> 5);

    const RowIds<T> ids = row_ids<T>(p, lane);
    const int lane_tyx = pack_tyx(p, lane < p.N ? lane : 0);
    RowRegs<T> rq, rk, rv, ro;
    int rid_next = 0;  // (as in k_attn_fwd)
    auto prefetch = [&](int64_t w) __attribute__((always_inline)) {
        int off[AC<T>::VPR], offo[AC<T>::VPR];
        const WinPos q = win_pos(p, w);
        row_offsets<T>(off, p, q, ids, C3, lane);
        row_offsets<T>(offo, p, q, ids, p.C, lane);
        const T* wb = qkv + q.base * C3;  // wave-uniform
        load_rows<T>(rq, wb, off);
        load_rows<T>(rk, wb + p.C, off);
        load_rows<T>(rv, wb + 2 * p.C, off);
        load_rows<T>(ro, dout + q.base * p.C, offo);
        rid_next = (p.mask_ids && lane < p.N) ? p.mask_ids[(int)(w % nWimg) * p.N + lane] : 0;
    };
    if (g < p.n_windows) prefetch(g);
    for (int64_t w = g; w < p.n_windows; w += p.G) {
        const int wm = (int)(w % nWimg);
        const WinPos wq = win_pos(p, w);
        __syncthreads();
        tok[lane] = lane < p.N ? token_at(p, wq, lane, lane_tyx) * C3 : 0;  // element offset of the token's dqkv row
        srid[lane] = rid_next;
        const bool masked = __ballot(lane < p.N && rid_next != __builtin_amdgcn_readfirstlane(rid_next)) != 0ull;  // (as in k_attn_fwd)
        T* gb = dqkv + wq.base * C3;  // wave-uniform
        store_rows<T>(sQ, rq, p.N, lane);
        store_rows<T>(sK, rk, p.N, lane);
        store_rows<T>(sV, rv, p.N, lane);
        store_rows<T>(sO, ro, p.N, lane);
        __syncthreads();
        if (w + p.G < p.n_windows) prefetch(w + p.G);

        f32x16 dk[2], dv[2];  // [sj]: dK^T / dV^T [d][key], accumulated over the query halves
        zero(dk[0]);
        zero(dk[1]);
        zero(dv[0]);
        zero(dv[1]);
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            if (si * 32 >= p.N) break;
            f32x16 pt[2];
            scores_t<T, DENSE>(pt, sK, sQ, si, p, B, srid, masked, wm, lane);
            (void)softmax_t<true, sizeof(T) == 4>(pt, B.c);
            // dP^T[j][i] = sum_d V[j][d] dO[i][d]
            f32x16 dp[2];
            zero(dp[0]);
            zero(dp[1]);
#pragma unroll
            for (int kt = 0; kt < AC<T>::KT_D; ++kt) {
                Frag<T> fo = rowfrag<T>(sO, si, kt, lane, p.N);
#pragma unroll
                for (int sj = 0; sj < 2; ++sj) {
                    Frag<T> fv = rowfrag<T>(sV, sj, kt, lane, p.N);
                    mtl_mma(fv, fo, dp[sj]);
                }
            }
            float D = 0.f;
#pragma unroll
            for (int sj = 0; sj < 2; ++sj)
#pragma unroll
                for (int r = 0; r < 16; ++r) D += pt[sj][r] * dp[sj][r];
            D += __shfl_xor(D, 32);
            // dS^T in place of dp; dbias accumulates in registers (padded keys have P = 0, padded queries are never written)
#pragma unroll
            for (int sj = 0; sj < 2; ++sj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ds = pt[sj][r] * (dp[sj][r] - D);
                    dp[sj][r] = ds;
                    dbacc[si][sj][r] += ds;
                }
            // hand P^T / dS^T of this half over for the key-owned products (previous half's reads are complete: the
            // wave executes in order and the MFMAs below consumed their fragments)
            __builtin_amdgcn_wave_barrier();
            img_store<T>(sPi, pt, lane);
            img_store<T>(sDi, dp, lane);
            // dQ^T[d][i] = scale * sum_j K[j][d] dS^T[j][i]
            f32x16 dq;
            zero(dq);
#pragma unroll
            for (int kt = 0; kt < AC<T>::KT_N; ++kt) {
                Frag<T> fk = colfrag(sK, kt, lane, p.N, (T*)nullptr);
                Frag<T> fs = regfrag(dp, kt, (T*)nullptr);
                mtl_mma(fk, fs, dq);
            }
            store_dt_lds<T>(stg, gb, tok + si * 32, p.N - si * 32, dq, p.scale, lane);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image stores have landed (single wave: no barrier needed)
            __builtin_amdgcn_wave_barrier();
            // dK^T[d][j] += sum_{i in half} Q[i][d] dS[i][j],  dV^T[d][j] += sum_i dO[i][d] P[i][j]
            // (padded queries: their Q / dO rows are the shared zero row, so whatever P / dS hold there drops out)
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                Frag<T> fq = colfrag(sQ, si * KQ + kq, lane, p.N, (T*)nullptr);
                Frag<T> fo = colfrag(sO, si * KQ + kq, lane, p.N, (T*)nullptr);
#pragma unroll
                for (int sj = 0; sj < 2; ++sj) {
                    Frag<T> fds = imgfrag(sDi, sj * 32, kq, lane, (T*)nullptr);
                    Frag<T> fp = imgfrag(sPi, sj * 32, kq, lane, (T*)nullptr);
                    mtl_mma(fq, fds, dk[sj]);
                    mtl_mma(fo, fp, dv[sj]);
                }
            }
        }
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
            if (sj * 32 < p.N) {
                store_dt_lds<T>(stg, gb + p.C, tok + sj * 32, p.N - sj * 32, dk[sj], p.scale, lane);
                store_dt_lds<T>(stg, gb + 2 * p.C, tok + sj * 32, p.N - sj * 32, dv[sj], 1.f, lane);
            }
        }
    }
    float* dst = p.dbias_part + ((int64_t)g * p.nH + head) * p.N * p.N;  // [j][i]
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int sj = 0; sj < 2; ++sj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = si * 32 + (lane & 31), j = sj * 32 + (r & 3) + 8 * (r >> 2) + h4;
                if (i < p.N && j < p.N) dst[j * p.N + i] = dbacc[si][sj][r];
            }
}

// dbias[h][i][j] = sum_g part[g][h][j][i]: one workgroup per 64 outputs, 4 waves stride over g, fixed-order combine
__global__ __launch_bounds__(256) void k_dbias_reduce(const float* part, float* dbias, int G, int nH, int N) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = nH * N * N;
    const int idx = blockIdx.x * 64 + lane;  // index in the [h][j][i] partial layout
    float a0 = 0.f, a1 = 0.f;
    if (idx < total) {
        int g = wave;
        for (; g + 4 < G; g += 8) {
            a0 += part[(int64_t)g * total + idx];
            a1 += part[(int64_t)(g + 4) * total + idx];
        }
        for (; g < G; g += 4) a0 += part[(int64_t)g * total + idx];
    }
    sm[wave][lane] = a0 + a1;
    __syncthreads();
    if (wave == 0 && idx < total) {
        const int h = idx / (N * N), rem = idx % (N * N);
        const int j = rem / N, i = rem % N;
        dbias[((int64_t)h * N + i) * N + j] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
    }
}

int check(const mtlora_attn_desc* d) {
    if (!d) return MTLORA_ERR_NULL;
    if (d->dtype != MTLORA_F32 && d->dtype != MTLORA_BF16 && d->dtype != MTLORA_F16) return MTLORA_ERR_DTYPE;
    if (d->head_dim != HD) return MTLORA_ERR_UNSUPPORTED;
    if (d->window_size <= 0 || d->window_size * d->window_size > AN) return MTLORA_ERR_UNSUPPORTED;
    if (d->B < 0 || d->H <= 0 || d->W <= 0 || d->num_heads <= 0) return MTLORA_ERR_SHAPE;
    if (d->H % d->window_size || d->W % d->window_size) return MTLORA_ERR_SHAPE;
    if (d->shift < 0 || d->shift >= d->window_size) return MTLORA_ERR_SHAPE;
    if (d->B * d->H * d->W >= ((int64_t)1 << 31)) return MTLORA_ERR_SHAPE;
    if ((int64_t)d->H * d->W * 3 * d->num_heads * d->head_dim * 4 >= ((int64_t)1 << 31)) return MTLORA_ERR_SHAPE;  // 32-bit in-image offsets
    return MTLORA_OK;
}

// persistent launch: G window-groups per head such that G * nH workgroups are all resident
// (LDS-limited workgroups per CU, capped by `cap`), never more groups than windows
static int groups_for(size_t lds_bytes, int cap, int nH, int64_t n_windows) {
    int per_cu = (int)((160 * 1024) / (lds_bytes + 512));
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    int64_t G = ((int64_t)256 * per_cu) / nH;
    if (G > n_windows) G = n_windows;
    if (G < 1) G = 1;
    return (int)G;
}

static size_t bwd_lds_bytes(const mtlora_attn_desc* d) {
    const int N = d->window_size * d->window_size;
    const int rs = d->dtype == MTLORA_F32 ? AC<float>::RS : AC<bf16>::RS;
    const size_t ib = (size_t)((img_rows_h(N) * rs + 15) / 16) * 16;
    const size_t img = d->dtype == MTLORA_F32 ? IMG<float>::RS : IMG<bf16>::RS;  // P / dS hand-over images
    return 4 * ib + 2 * 32 * img + 2 * AN * 4 + (d->dtype == MTLORA_F32 ? STG<float>::BYTES : STG<bf16>::BYTES) + (d->dtype == MTLORA_F32 ? (size_t)N * bias_stride_c(N) * 4 : 0);  // (fp32: bias image)
}

int bwd_groups(const mtlora_attn_desc* d) {
    const int64_t nwin = d->B * (d->H / d->window_size) * (d->W / d->window_size);
    // MTL_ATTN_BWD_WPS single-wave workgroups per SIMD (the register budget the kernel is compiled for), LDS permitting
    return groups_for(bwd_lds_bytes(d), 4 * MTL_ATTN_BWD_WPS, d->num_heads, nwin);
}

AttnParams make_params(const mtlora_attn_desc* d) {
    AttnParams p = {};
    p.H = d->H;
    p.W = d->W;
    p.ws = d->window_size;
    p.shift = d->shift;
    p.nH = d->num_heads;
    p.N = d->window_size * d->window_size;
    p.C = d->num_heads * d->head_dim;
    p.nWx = d->W / d->window_size;
    p.nWy = d->H / d->window_size;
    p.n_windows = d->B * p.nWx * p.nWy;
    p.image_layout = d->image_layout;
    p.scale = d->scale;
    return p;
}

}  // namespace

extern "C" {

int64_t mtlora_window_attn_bwd_scratch_bytes(const mtlora_attn_desc* d) {
    if (check(d) != MTLORA_OK) return -1;
    const int N = d->window_size * d->window_size;
    return (int64_t)bwd_groups(d) * d->num_heads * N * N * 4 + 256;
}

int mtlora_window_attn_fwd(const mtlora_attn_desc* d, const void* qkv, const float* bias, const float* mask,
                           const int32_t* mask_ids, void* out, void* stream) {
    int st = check(d);
    if (st != MTLORA_OK) return st;
    if (!qkv || !bias || !out) return MTLORA_ERR_NULL;
    if (((uintptr_t)qkv | (uintptr_t)out) & 15u) return MTLORA_ERR_ALIGN;
    AttnParams p = make_params(d);
    if (p.n_windows == 0) return MTLORA_OK;
    p.qkv = qkv;
    p.bias = bias;
    p.mask = mask;
    p.mask_ids = mask_ids;
    p.mask_value = d->mask_value;
    p.out = out;
    const int rs = d->dtype == MTLORA_F32 ? AC<float>::RS : AC<bf16>::RS;
    const size_t ib = (size_t)((img_rows_h(p.N) * rs + 15) / 16) * 16;
    const size_t lds = 3 * ib + 2 * AN * 4 + (d->dtype == MTLORA_F32 ? STG<float>::BYTES : STG<bf16>::BYTES) + (d->dtype == MTLORA_F32 ? (size_t)p.N * bias_stride_c(p.N) * 4 : 0);  // (fp32: bias image)
    p.G = groups_for(lds, 4 * MTL_ATTN_FWD_WPS, p.nH, p.n_windows);  // persistent grid: exactly the resident workgroups
    const unsigned grid = (unsigned)(p.G * p.nH);
    hipStream_t s = (hipStream_t)stream;
    const double ab = 4.0 * mtl_elem_size(d->dtype) * (double)p.n_windows * p.N * p.C;
    MtlProfScope prof(PK_ATTN_FWD, ab, s, ab, 4.0 * (double)p.n_windows * p.N * p.N * p.C);
    const bool dense = p.mask && !p.mask_ids;
    if (d->dtype == MTLORA_F32) {
        if (dense)
            hipLaunchKernelGGL((k_attn_fwd<float, true>), dim3(grid), dim3(64), lds, s, p);
        else
            hipLaunchKernelGGL((k_attn_fwd<float, false>), dim3(grid), dim3(64), lds, s, p);
    } else if (d->dtype == MTLORA_F16) {
        if (dense)
            hipLaunchKernelGGL((k_attn_fwd<f16, true>), dim3(grid), dim3(64), lds, s, p);
        else
            hipLaunchKernelGGL((k_attn_fwd<f16, false>), dim3(grid), dim3(64), lds, s, p);
    } else {
        if (dense)
            hipLaunchKernelGGL((k_attn_fwd<bf16, true>), dim3(grid), dim3(64), lds, s, p);
        else
            hipLaunchKernelGGL((k_attn_fwd<bf16, false>), dim3(grid), dim3(64), lds, s, p);
    }
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_window_attn_bwd(const mtlora_attn_desc* d, const void* qkv, const float* bias, const float* mask,
                           const int32_t* mask_ids, const void* dout, void* dqkv, float* dbias, void* scratch,
                           int64_t scratch_bytes, void* stream) {
    int st = check(d);
    if (st != MTLORA_OK) return st;
    if (!qkv || !bias || !dout || !dqkv || !dbias || !scratch) return MTLORA_ERR_NULL;
    if (((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv | (uintptr_t)scratch) & 15u) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_window_attn_bwd_scratch_bytes(d) - 256) return MTLORA_ERR_WORKSPACE;
    AttnParams p = make_params(d);
    hipStream_t s = (hipStream_t)stream;
    if (p.n_windows == 0) {
        mtl_zero_async(dbias, (size_t)p.nH * p.N * p.N * 4, s);
        return MTLORA_OK;
    }
    p.qkv = qkv;
    p.bias = bias;
    p.mask = mask;
    p.mask_ids = mask_ids;
    p.mask_value = d->mask_value;
    p.dout = dout;
    p.dqkv = dqkv;
    p.dbias_part = reinterpret_cast<float*>(scratch);
    p.G = bwd_groups(d);
    const unsigned grid = (unsigned)(p.G * p.nH);
    const size_t lds = bwd_lds_bytes(d);
    {
        const double ab = 7.0 * mtl_elem_size(d->dtype) * (double)p.n_windows * p.N * p.C;
        MtlProfScope prof(PK_ATTN_BWD, ab, s, ab, 8.0 * (double)p.n_windows * p.N * p.N * p.C);
        const bool dense = p.mask && !p.mask_ids;
        if (d->dtype == MTLORA_F32) {
            if (dense)
                hipLaunchKernelGGL((k_attn_bwd<float, true>), dim3(grid), dim3(64), lds, s, p);
            else
                hipLaunchKernelGGL((k_attn_bwd<float, false>), dim3(grid), dim3(64), lds, s, p);
        } else if (d->dtype == MTLORA_F16) {
            if (dense)
                hipLaunchKernelGGL((k_attn_bwd<f16, true>), dim3(grid), dim3(64), lds, s, p);
            else
                hipLaunchKernelGGL((k_attn_bwd<f16, false>), dim3(grid), dim3(64), lds, s, p);
        } else {
            if (dense)
                hipLaunchKernelGGL((k_attn_bwd<bf16, true>), dim3(grid), dim3(64), lds, s, p);
            else
                hipLaunchKernelGGL((k_attn_bwd<bf16, false>), dim3(grid), dim3(64), lds, s, p);
        }
    }
    const int total = p.nH * p.N * p.N;
    hipLaunchKernelGGL(k_dbias_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, (const float*)p.dbias_part,
                       dbias, p.G, p.nH, p.N);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}
}
