// linear.hip -- MTLoRALinear forward / backward on CDNA4 (gfx950).
// Replaces the ATen sequence of models/lora.py:253-284 and its autograd backward.
//
// Kernel families (all hand-written MFMA, 64-wide waves):
//
//   k_pack   fp32 LoRA masters -> compute-dtype packed factors: A_cat (R x K), B_cat (N x R), At_cat (K x R), Bt_cat (R x N),
//              alpha (R), the alpha-scaled projection rows a_proj / bt_proj and the fragment-major, k-permuted expansion
//              factors b_frag / at_frag of the wave-streaming kernels;   R = sum_o rp(o), every rank padded to 8 columns
//   k_sp_*   (stream.h) WAVE-STREAMING kernels, the default wherever a launch is eligible (16-bit types, stationary operands
//            fit in LDS): k_sp_xres / k_sp_ares = ONE launch per T = 0 layer and direction (projection kept in registers, no
//            P / Q pass, no re-read of X / dY; k_sp_xres also carries the GELU second output and the GELU' gate), k_sp_proj =
//            the P / Q passes of every other layer, k_sp_projsum = Q of all outputs + G = sum of the gradients in one pass
//            (T <= 4), k_sp_projk = P / Q with projection rows too large for LDS (single-round launches), k_sp_tn = the
//            factor gradients dA / dB with transposed LDS reads (large row counts)
//   k_ntd    (dense.h) single-output launches with a long reduction and well-filled residency rounds: 256 x 128 x 64 tiles,
//            global -> LDS ring of three stages running across the tiles of a persistent workgroup
//   k_rank_out  the task outputs of a dX launch with small task ranks (dX_t = Q_t A_t [.* gelu'(h_t)]) as a streaming kernel
//   k_nt     "NT" tile GEMM  D[n][m] = sum_k Wgt[n][k] * Act[m][k]  (128 x 128 tile, 4 or 8 waves per workgroup) run
//            as ONE software-pipelined stream of k-tiles over the base GEMM and every output's rank segment, with
//              - multi-source activation (sum of up to 1+T tensors formed while staging: G = dY_s + sum dY_t)
//              - optional dropout mask applied to the staged activation at LDS-store time (P = alpha * D(X) A^T)
//              - bias / per-row alpha epilogue, GELU second output (ACT), GELU' gate (GATE)
//              - multi-output low-rank parts: for each output o, extra k-tiles over the o-th rank segment of
//                L (M x R) and R (N x R) chained onto the base accumulator (Y_o = base + L_o R_o^T), optionally
//                masked (dX = G W + keep .* (Q A))
//            The MFMA "A" operand is the weight tile and "B" the activation tile, so a lane's four consecutive
//            accumulator registers are four consecutive output COLUMNS; the bf16 epilogue goes through LDS so that
//            stores are whole 128-byte row segments.   k_ntl = its single-output bf16 launches as straight-line code.
//   k_tn     "TN" split-M reduction  Out[a][b] = sum_m SrcA[m][a] * SrcB[m][b]  (dA = Q^T D(X), dB^T = P^T dY) with
//            64 (rank side) x 256 (wide side) tiles: both operands are read with the LDS transpose load
//            (ds_read_b64_tr_b16) for bf16; per-split partials (deterministic) + k_tn_reduce.
//   k_sum    G = sum of the output gradients (matrixv2 factors; pre-summed dX operand of wide outputs).
//
// Forward  = k_pack, then  k_sp_xres (T = 0, K <= 192)  |  k_sp_proj / k_sp_projk / k_nt (P), k_ntd / k_ntl / k_nt (all 1+T outputs).
// Backward = T = 0: k_sp_ares (narrow input) / k_sp_xres (wide input, short reduction, gate): Q + dX together
//            | else k_sp_projsum (Q + G, T <= 4) / [k_sum] k_sp_proj / k_sp_projk / k_nt (Q), then k_ntd / k_ntl / k_nt (dX) [+ k_rank_out (dX_t)];
//            k_sp_tn / k_tn + reduce for dA / dB.
// DESIGN.md section 4.1 / 4.3 has the measurements and the experiments that were tried and dropped.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.h"
#include "internal.h"
#include "hid_params.h"

namespace {

constexpr int TILE = 128;    // rows per CTA tile, both operands
constexpr int SUBT = 3;             // 64-byte MFMA sub-tiles per staged k-tile: K = 96 bf16 is ONE round trip
constexpr int VPT = SUBT;           // 16-byte vectors per thread per tile row
constexpr int ROWB = 64 * SUBT;     // payload bytes per row per k-tile (96 bf16 / 48 f32)
constexpr int LDSB = ROWB + 16;     // padded LDS row stride (conflict-free ds_read_b128, 16-B aligned)
constexpr int EPI_ROW = 64 * 2 + 8;                      // row stride of a wave's bf16 output image (epilogue)
constexpr int EPI_BYTES = 4 * 64 * EPI_ROW;               // 4 waves x 64 rows = 8 waves x 32 rows
constexpr int STAGE_BYTES = 2 * TILE * LDSB > EPI_BYTES ? 2 * TILE * LDSB : EPI_BYTES;  // staging / epilogue region
constexpr int MAXO = MTLORA_MAX_TASKS + 1;
// MTLORA_NT_DBG ablation toggles (tools/nt_ablate.sh) are compiled in only with -DMTL_NT_ABLATE=1 (MTLORA_ABLATE=1 for
// csrc/build.py): the runtime tests cost the generic kernels ~10 SGPR spills each
#ifndef MTL_NT_ABLATE
#define MTL_NT_ABLATE 0
#endif
constexpr int NT_DBG_MASK = MTL_NT_ABLATE ? ~0 : 0;

// ------------------------------------------------------------------------------------------------
// segment table: output o (0 = shared, 1..T = tasks) owns columns [off, off + rp) of the rank axis
// ------------------------------------------------------------------------------------------------
struct Segs {
    int n;  // 1 + T
    int r[MAXO], rp[MAXO], off[MAXO];
    int R;     // row stride of P / Q / the packed factors (columns), a multiple of 16
    int used;  // columns that belong to a segment: [used, R) is padding nobody writes
};

static Segs make_segs(const mtlora_linear_desc* d) {
    Segs s;
    s.n = 1 + d->T;
    int off = 0;
    for (int o = 0; o < s.n; ++o) {
        int r = (o == 0) ? d->r_s : d->r_t[o - 1];
        s.r[o] = r;
        // a segment is padded to ONE 16-byte vector of bf16 (8 columns), not to the 16-wide MFMA k granule: every consumer masks
        // per 16-byte vector (k_nt zero-fills the vectors past a part's k range on both operands, k_tn windows start on vector
        // boundaries), so r_t = 4 costs 8 columns of P / Q / factors instead of 16 -- R = 96 instead of 128 at C2's task layers,
        // 80 instead of 144 with 8 tasks of rank 4.  (Packing two 4-wide segments into one vector would need the P pass to merge
        // two different activation sources into one store: DESIGN.md 7.)
        s.rp[o] = (int)mtl_round_up(r, 8);
        s.off[o] = off;
        off += s.rp[o];
    }
    for (int o = s.n; o < MAXO; ++o) s.r[o] = s.rp[o] = s.off[o] = 0;
    s.used = off;
    s.R = (int)mtl_round_up(off, 16);  // row stride of P / Q: whole 32-byte pairs (the trailing pad columns are never read)
    return s;
}

// packed factors (offsets relative to the pack base: the head of the ctx buffer, or the caller's persistent d->packed buffer) and
// P (offset relative to the ctx base)
struct CtxLayout {
    int64_t a_cat, b_cat, at_cat, bt_cat, alpha, a_proj, bt_proj, b_frag, at_frag, pack_total, p, total;
};
static CtxLayout ctx_layout(const mtlora_linear_desc* d, const Segs& s) {
    const int es = mtl_elem_size(d->dtype);
    CtxLayout L;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        int64_t at = o;
        o += mtl_round_up(bytes, 256);
        return at;
    };
    L.a_cat = take((int64_t)s.R * d->K * es);
    L.b_cat = take((int64_t)d->N * s.R * es);
    L.at_cat = take((int64_t)d->K * s.R * es);
    L.bt_cat = take((int64_t)s.R * d->N * es);
    L.alpha = take((int64_t)s.R * 4);
    L.a_proj = take((int64_t)s.R * d->K * es);   // alpha * A_cat   (projection rows of the wave-streaming forward / P pass)
    L.bt_proj = take((int64_t)s.R * d->N * es);  // alpha * Bt_cat  (projection rows of the wave-streaming dX / Q pass)
    // expansion factors of the wave-streaming kernels (stream.h), FRAGMENT-major and k-permuted: fragment (32-row block b,
    // 16-wide rank step t) = 1 KB, lane l = (row b*32 + (l & 31), h = l >> 5) holds the 8 rank columns
    // 16 t + 8 (s >> 2) + 4 h + (s & 3), s = 0..7 -- the order in which a lane holds P^T / Q^T after the projection MFMA
    // (rank steps padded to whole 32-row projection blocks: 2 * ceil(R / 32) steps per block, zero past the segments)
    L.b_frag = take(mtl_round_up(d->N, 32) * mtl_round_up(s.R, 32) * es);   // rows = output columns n:  B_cat[n][r]
    L.at_frag = take(mtl_round_up(d->K, 32) * mtl_round_up(s.R, 32) * es);  // rows = input columns k:   A_cat[r][k]
    L.pack_total = o;
    if (d->packed) o = 0;  // the factors live in the caller's buffer: ctx holds P alone
    L.p = take(d->M * s.R * es);
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
// k_pack
// ------------------------------------------------------------------------------------------------
struct PackParams {
    const float* A[MAXO];
    const float* B[MAXO];
    float alpha[MAXO];
    Segs s;
    int K, N;
};

// The packing walks 64 x 64 TILES of the two concatenated factor matrices -- A_cat (R x K, rows = rank columns rr) and B_cat (N x R) -- one
// tile per workgroup pass: the fp32 masters are read along their contiguous axis (K for A, the segment's rank for B), the same-orientation
// copies (a_cat / a_proj, b_cat) are stored from registers, and the tile goes through LDS once for everything that is transposed or
// permuted (at_cat, bt_cat / bt_proj, the fragment-major at_frag / b_frag): every global access of the launch is a coalesced row segment.
// (Round 3's element-wise walk scattered 2-byte stores with a stride of R for the transposed copies and divided 64-bit indices per
// element: 1.07 ms for the 72 layers of Swin-B at rank 128, 30 x the time of the bytes it moves.)
constexpr int PK_T = 64;

template <typename PP>
__device__ __forceinline__ int pack_seg_of(const PP& p, int rr) {
    int o = 0;
#pragma unroll
    for (int q = 1; q < MAXO; ++q)
        if (q < p.s.n && rr >= p.s.off[q]) o = q;
    return o;
}

template <typename T, typename PP>
__device__ __forceinline__ void pack_body(const PP& p, T* a_cat, T* b_cat, T* at_cat, T* bt_cat, float* alpha, T* a_proj, T* bt_proj, T* b_frag,
                                          T* at_frag, int bid, int nblk) {
    __shared__ float tile[PK_T][PK_T + 1];
    constexpr bool FRAG = sizeof(T) == 2;  // fragment-major expansion factors: 16-bit types only (the wave-streaming kernels)
    const int R = p.s.R, K = p.K, N = p.N;
    const int tr = (R + PK_T - 1) / PK_T, tk = (K + PK_T - 1) / PK_T, tn = (N + PK_T - 1) / PK_T;
    const int na_t = tr * tk, nb_t = tn * tr;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int R16 = ((R + 31) >> 5) << 1, K32 = (K + 31) & ~31, N32 = (N + 31) & ~31;
    for (int job = bid; job < na_t + nb_t; job += nblk) {
        const bool isa = job < na_t;
        const int jb = isa ? job : job - na_t;
        // tile origin: rows r0 / cols c0 of A_cat (rr, k), or rows n0 / cols r0 of B_cat (n, rr)
        const int row0 = isa ? (jb / tk) * PK_T : (jb / tr) * PK_T;
        const int col0 = isa ? (jb % tk) * PK_T : (jb % tr) * PK_T;
        float v[PK_T / 4];  // all 16 loads of the tile are in flight before the first store (the stores may alias for all the compiler knows)
        if (isa) {
            const int c = col0 + tx;
            float al[PK_T / 4];
#pragma unroll
            for (int i = 0; i < PK_T / 4; ++i) {
                const int rr = row0 + ty + 4 * i;  // (wave-uniform)
                const int o = pack_seg_of(p, rr), lr = rr - p.s.off[o];
                al[i] = p.alpha[o];
                v[i] = (rr < R && lr < p.s.r[o] && c < K && p.A[o]) ? p.A[o][(int64_t)lr * K + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < PK_T / 4; ++i) {
                const int row = ty + 4 * i, rr = row0 + row;
                tile[row][tx] = v[i];
                if (rr < R && c < K) {
                    a_cat[(int64_t)rr * K + c] = mtl_from_f32<T>(v[i]);
                    a_proj[(int64_t)rr * K + c] = mtl_from_f32<T>(v[i] * al[i]);
                }
            }
            if (col0 == 0 && threadIdx.x < PK_T && row0 + tx < R) alpha[row0 + tx] = p.alpha[pack_seg_of(p, row0 + tx)];
        } else {
            const int rr = col0 + tx;
            const int o = pack_seg_of(p, rr), lr = rr - p.s.off[o], ro = p.s.r[o];
            const bool live = rr < R && lr < ro && p.B[o];
            const float* __restrict__ Bo = p.B[o];
#pragma unroll
            for (int i = 0; i < PK_T / 4; ++i) {
                const int n = row0 + ty + 4 * i;
                v[i] = (live && n < N) ? Bo[(int64_t)n * ro + lr] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < PK_T / 4; ++i) {
                const int row = ty + 4 * i, n = row0 + row;
                tile[row][tx] = v[i];
                if (rr < R && n < N) b_cat[(int64_t)n * R + rr] = mtl_from_f32<T>(v[i]);
            }
        }
        __syncthreads();
        if (isa) {  // at_cat[k][rr]: lanes along rr
            const int rr = row0 + tx;
#pragma unroll 4
            for (int i = 0; i < PK_T / 4; ++i) {
                const int cl = ty + 4 * i, c = col0 + cl;
                if (rr < R && c < K) at_cat[(int64_t)c * R + rr] = mtl_from_f32<T>(tile[tx][cl]);
            }
        } else {    // bt_cat / bt_proj[rr][n]: lanes along n
            const int n = row0 + tx;
#pragma unroll 4
            for (int i = 0; i < PK_T / 4; ++i) {
                const int rl = ty + 4 * i, rr = col0 + rl;  // (wave-uniform)
                if (rr < R && n < N) {
                    const float v = tile[tx][rl];
                    bt_cat[(int64_t)rr * N + n] = mtl_from_f32<T>(v);
                    bt_proj[(int64_t)rr * N + n] = mtl_from_f32<T>(v * p.alpha[pack_seg_of(p, rr)]);
                }
            }
        }
        if constexpr (FRAG) {
            // fragment (32-row block blk of k or n, 16-wide rank step t) = 512 elements: lane l = (row blk * 32 + (l & 31), h = l >> 5)
            // holds rank columns 16 t + 8 (s >> 2) + 4 h + (s & 3), s = 0..7.  The tile holds 2 x 4 fragments; a thread writes two
            // (fragment, lane) groups of 8 elements = one 16-byte store each.
            T* __restrict__ dst = isa ? at_frag : b_frag;
            const int blk0 = (isa ? col0 : row0) >> 5, t0 = (isa ? row0 : col0) >> 4, nblk32 = (isa ? K32 : N32) >> 5;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int q = threadIdx.x + 256 * h2, fl = q >> 6, ln = q & 63;
                const int bl = fl >> 2, tl = fl & 3, rl = bl * 32 + (ln & 31);
                if (blk0 + bl < nblk32 && t0 + tl < R16) {
                    T tmp[8];
#pragma unroll
                    for (int sidx = 0; sidx < 8; ++sidx) {
                        const int kl = 16 * tl + 8 * (sidx >> 2) + 4 * (ln >> 5) + (sidx & 3);
                        tmp[sidx] = mtl_from_f32<T>(isa ? tile[kl][rl] : tile[rl][kl]);
                    }
                    const int64_t f = (int64_t)(blk0 + bl) * R16 + (t0 + tl);
                    *reinterpret_cast<uint4*>(dst + f * 512 + ln * 8) = *reinterpret_cast<const uint4*>(tmp);
                }
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pack(PackParams p, T* a_cat, T* b_cat, T* at_cat, T* bt_cat, float* alpha, T* a_proj,
                                              T* bt_proj, T* b_frag, T* at_frag) {
    pack_body<T>(p, a_cat, b_cat, at_cat, bt_cat, alpha, a_proj, bt_proj, b_frag, at_frag, (int)blockIdx.x, (int)gridDim.x);
}

// one launch for EVERY layer of a model (mtlora_linear_pack_table): blockIdx.y = table entry.  The factors change once per optimizer
// step, so a trainer refreshes all the packed buffers here instead of paying one k_pack launch per layer and forward call
// (48 launches per step at C2, 96 at C4).
struct PackEntry {
    PackParams pp;
    unsigned char* dst;  // the layer's packed buffer (device)
    int64_t off[9];      // a_cat, b_cat, at_cat, bt_cat, alpha, a_proj, bt_proj, b_frag, at_frag
};
template <typename T>
__global__ __launch_bounds__(256) void k_pack_table(const PackEntry* __restrict__ table) {
    const PackEntry& e = table[blockIdx.y];
    unsigned char* d = e.dst;
    pack_body<T>(e.pp, reinterpret_cast<T*>(d + e.off[0]), reinterpret_cast<T*>(d + e.off[1]), reinterpret_cast<T*>(d + e.off[2]),
                 reinterpret_cast<T*>(d + e.off[3]), reinterpret_cast<float*>(d + e.off[4]), reinterpret_cast<T*>(d + e.off[5]),
                 reinterpret_cast<T*>(d + e.off[6]), reinterpret_cast<T*>(d + e.off[7]), reinterpret_cast<T*>(d + e.off[8]), (int)blockIdx.x,
                 (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// k_nt
// ------------------------------------------------------------------------------------------------
struct NtOut {
    void* ptr;       // (M x ld_out) output, element type T
    int seg_lo, seg_hi;  // rank-column range of L/R chained onto this output ([lo,hi) empty -> none)
    int use_base;    // add the shared base accumulator
    int mask_lr;     // multiply the low-rank part by the dropout keep mask of (m, n)
    int fold;        // after storing: base += low-rank part ('matrixv2': tasks see the shared update)
    const void* gate;  // GATE kernels: out *= gelu'(gate[m][n]) (same shape / dtype / row stride as the output), nullable
    void* act;         // ACT kernels: second output gelu(out) (same shape / dtype / row stride), nullable
};

struct NtParams {
    // base GEMM
    const void* act[MAXO];  // (M x K) sources, summed while staging
    int n_act;
    int act_mask;           // dropout keep-mask applied to the staged activation
    int64_t ld_act;
    const void* wgt;        // (n_rows x K)
    int64_t ld_wgt;
    int64_t M;
    int n_rows;             // rows of wgt == output columns
    int K;                  // reduction length of the base GEMM (0 -> no base GEMM)
    const float* bias;      // per output column, nullable
    const float* alpha;     // per output column multiplier on the base GEMM, nullable
    // low-rank epilogue
    const void* L;          // (M x ldL)
    const void* Rm;         // (n_rows x ldR)
    int64_t ldL, ldR;
    int n_out;
    NtOut out[MAXO];
    int64_t ld_out;
    // batched form (gridDim.z = nz > 0): z selects activation / weight row slab / output column slab
    int nz;
    const void* zact[MAXO];
    int zrow0[MAXO], zrows[MAXO], zmask[MAXO];
    int dbg;               // MTLORA_NT_DBG ablation bits (tools only): 1 no global stores, 2 no global loads, 4 no MFMA, 8 no epilogue
    DropoutCfg drop;
};

// kernel parameters are read straight from the kernarg segment (constant address space): dynamic indexing of a
// by-value struct argument would make the compiler copy the whole struct to scratch
typedef const __attribute__((address_space(4))) NtParams* NtPtr;

// RI = tile rows per thread per operand: 2 with 256 threads (4 waves), 1 with 512 threads (8 waves)
// A thread stages 3 * RI 16-byte vectors per operand per k-tile.  Which (row, vector) a thread owns is chosen for the LDS
// STORE: ds_write_b128 is served 8 lanes (128 bytes = all 32 banks) at a time, so 8 consecutive lanes must write 8 pieces that
// are distinct mod 128 bytes.  With the 208-byte row stride (13 pieces: odd, which is what keeps the ds_read_b128 fragment
// reads conflict-free) the earlier 4-lanes-per-row map put (row r, piece 0) and (row r+1, piece 3) on the same banks in
// every group -- a 2-way conflict on every staging store (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25,
// profiles/r01_pmc_sq.csv).  Now: slots 0 .. 2 RI - 1: 8 lanes x 16 B = the first 128 bytes of ONE row (also a 128-byte
// global-load segment instead of two 64-byte ones); the last RI slots: the remaining 64 bytes of rows r and r + 4 (52 pieces
// apart = 4 mod 8: the two halves interleave).
// The 4-wave variants (RI = 2: MULTI / row-panel / f32, already at 256 VGPRs) keep the 4-lanes-per-row map: six distinct row
// addresses per thread pushed their hot loop into scratch (8 -> 104 bytes per lane for the multi-output forward).
template <int RI>
__device__ __forceinline__ void nt_map(int tid, int slot, int& row, int& vec) {
    constexpr int NT = 512 / RI;
    if constexpr (RI == 2) {
        row = (tid >> 2) + (slot & 1) * 64;
        vec = (tid & 3) + 4 * (slot >> 1);
    } else if (slot < 2 * RI) {
        const int idx = tid + NT * slot;  // 0 .. 1023
        row = idx >> 3;
        vec = idx & 7;
    } else {
        const int idx = tid + NT * (slot - 2 * RI);  // 0 .. 511
        const int g = idx >> 3, l = idx & 7;
        row = (g >> 2) * 8 + (g & 3) + 4 * (l >> 2);
        vec = 8 + (l & 3);
    }
}

template <typename T, int RI>
struct TileRegs {
    u32x4 w[3 * RI], a[3 * RI];
    int mask;  // dropout keep-mask still to be applied to a[] (done at LDS-store time: applying it at load time
    int k0;    // would put an s_waitcnt vmcnt(0) behind every single load and serialise the tile's loads)
};

// stage one 128-row x ROWB-byte k-tile of the weight-like and activation-like operands into registers
template <typename T, bool MS, int RI>
__device__ __forceinline__ void nt_load(TileRegs<T, RI>& rg, int tid, const T* wgt, int64_t ld_w, int w_row0, int w_rows,
                                        const void* act0, NtPtr P, int n_act, int64_t ld_a, int64_t a_row0,
                                        int64_t a_rows, int k0, int k_hi, bool mask, int w_lo = 0) {
    constexpr int VEC = ET<T>::VEC;
    rg.mask = mask ? 1 : 0;
    rg.k0 = k0;
#pragma unroll
    for (int sl = 0; sl < 3 * RI; ++sl) {
        int r, v;
        nt_map<RI>(tid, sl, r, v);
        const int k = k0 + v * VEC;
        const bool kin = k < k_hi;
        const int wr = w_row0 + r;
        const bool wok = wr < w_rows && wr >= w_lo;
        rg.w[sl] = (kin && wok) ? *reinterpret_cast<const u32x4*>(wgt + (int64_t)wr * ld_w + k) : u32x4{0u, 0u, 0u, 0u};
        const int64_t ar = a_row0 + r;
        const bool aok = ar < a_rows && act0 != nullptr;
        const int64_t aoff = ar * ld_a;
        if (kin && aok) {
            u32x4 x = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(act0) + aoff + k);
            if (MS && n_act > 1) {
                if constexpr (sizeof(T) == 4) {
                    f32x4 fx = __builtin_bit_cast(f32x4, x);
#pragma unroll
                    for (int s = 1; s < MAXO; ++s) {
                        if (s < n_act) fx += *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(P->act[s]) + aoff + k);
                    }
                    x = __builtin_bit_cast(u32x4, fx);
                } else {
                    float f[8];
                    VOps<T>::unpack(x, f);
#pragma unroll
                    for (int s = 1; s < MAXO; ++s) {
                        if (s < n_act) {
                            const u32x4 y = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(P->act[s]) + aoff + k);
                            float g[8];
                            VOps<T>::unpack(y, g);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g[e];
                        }
                    }
                    x = VOps<T>::pack(f);
                }
            }
            rg.a[sl] = x;
        } else {
            rg.a[sl] = u32x4{0u, 0u, 0u, 0u};
        }
    }
}

template <typename T, int RI>
__device__ __forceinline__ void nt_store_lds(TileRegs<T, RI>& rg, int tid, unsigned char* sW, unsigned char* sA,
                                             const DropoutCfg& dc, int64_t a_row0) {
    constexpr int VEC = ET<T>::VEC;
#pragma unroll
    for (int sl = 0; sl < 3 * RI; ++sl) {
        int r, v;
        nt_map<RI>(tid, sl, r, v);
        if (rg.mask) {  // wave-uniform
            const uint32_t rh = mtl_dropout_rowhash(dc, 0u, (uint32_t)(a_row0 + r));
            VOps<T>::drop(rg.a[sl], dc, rh, (uint32_t)(rg.k0 + v * VEC));
        }
        *reinterpret_cast<u32x4*>(sW + r * LDSB + v * 16) = rg.w[sl];
        *reinterpret_cast<u32x4*>(sA + r * LDSB + v * 16) = rg.a[sl];
    }
}

// multiply the staged tile; k_left = elements of the part's k range still ahead (sub-tiles past it are all zero
// and skipped -- wave-uniform)
// SM = 32-row m sub-blocks per wave: 2 (4 waves, wave tile 64 n x 64 m) or 1 (8 waves, wave tile 64 n x 32 m)
template <typename T, int SM>
__device__ __forceinline__ void nt_compute(f32x16 (&acc)[2][SM], const unsigned char* sW, const unsigned char* sA,
                                           int lane, int wn, int wm, int k_left, int a_stride = LDSB) {
    constexpr int KS = 64 / (int)sizeof(T);  // elements per 64-byte sub-tile
    const int h = lane >> 5, rl = lane & 31;
#pragma unroll
    for (int t = 0; t < SUBT; ++t) {
        if (t * KS >= k_left) break;
        Frag<T> fw[2], fa[SM];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char* pw = sW + (wn * 64 + s * 32 + rl) * LDSB + t * 64;
            fw[s].v[0] = *reinterpret_cast<const u32x4*>(pw + h * 16);
            fw[s].v[1] = *reinterpret_cast<const u32x4*>(pw + (2 + h) * 16);
        }
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            const unsigned char* pa = sA + (wm * (32 * SM) + s * 32 + rl) * a_stride + t * 64;
            fa[s].v[0] = *reinterpret_cast<const u32x4*>(pa + h * 16);
            fa[s].v[1] = *reinterpret_cast<const u32x4*>(pa + (2 + h) * 16);
        }
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
            for (int sm = 0; sm < SM; ++sm) mtl_mma(fw[sn], fa[sm], acc[sn][sm]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_nt: one software-pipelined TILE STREAM per workgroup.
// The k-tiles of the base GEMM and of every output's rank segment are consumed as ONE sequence: the register
// prefetch of tile i+1 is issued before tile i is multiplied, ACROSS the boundaries between base / outputs, so a
// workgroup pays the global->LDS latency once instead of once per output (with K = 96 a base GEMM is only
// 3 k-tiles; cold starts per output were the dominant cost).
//   MULTI  (forward with task outputs):  base | out0: base+lr0 | out1: base+lr1 ...   (base kept in registers)
//   lean   (everything else):            per output: lr_o -> [mask] -> base -> store   (ONE accumulator set:
//          the low-rank part is formed first so the dropout mask of dX = G W + keep.(Q A) applies to it alone)
// ------------------------------------------------------------------------------------------------
struct NtCursor {
    int q;         // index in the part sequence
    int k0;        // current k-tile origin
    int k_hi;      // end of the part's k range
    int lr;        // 1: rank-segment part (L x Rm), 0: base part (act x wgt)
    int valid;
    int bn;        // n-tile of the workgroup
};

__device__ __forceinline__ NtOut nt_out(NtPtr P, int o) {
    NtOut O;
    O.ptr = P->out[o].ptr;
    O.seg_lo = P->out[o].seg_lo;
    O.seg_hi = P->out[o].seg_hi;
    O.use_base = P->out[o].use_base;
    O.mask_lr = P->out[o].mask_lr;
    O.fold = P->out[o].fold;
    O.gate = P->out[o].gate;
    O.act = P->out[o].act;
    return O;
}

// k range of part q.  MULTI: q = 0 base, q = 1 + o rank segment of output o.
// lean: q = 2 o rank segment of output o, q = 2 o + 1 base (if that output uses it).
template <bool MULTI>
__device__ __forceinline__ void nt_part(NtPtr P, int q, int& lr, int& k_lo, int& k_hi) {
    if (MULTI) {
        if (q == 0) {
            lr = 0;
            k_lo = 0;
            k_hi = P->K;
        } else {
            const NtOut O = nt_out(P, q - 1);
            lr = 1;
            k_lo = O.seg_lo;
            k_hi = O.seg_hi;
        }
    } else {
        const NtOut O = nt_out(P, q >> 1);
        if (q & 1) {
            lr = 0;
            k_lo = 0;
            k_hi = O.use_base ? P->K : 0;
        } else {
            lr = 1;
            k_lo = O.seg_lo;
            k_hi = O.seg_hi;
        }
    }
}

template <bool MULTI>
__device__ __forceinline__ NtCursor nt_seek(NtPtr P, int q, int nseq, int bn) {
    NtCursor c;
    c.valid = 0;
    c.q = q;
    c.k0 = c.k_hi = c.lr = 0;
    c.bn = bn;
    for (; q < nseq; ++q) {
        int lr, lo, hi;
        nt_part<MULTI>(P, q, lr, lo, hi);
        if (hi > lo) {
            c.q = q;
            c.k0 = lo;
            c.k_hi = hi;
            c.lr = lr;
            c.valid = 1;
            return c;
        }
    }
    return c;
}

// MLR: some output masks its low-rank part (dX = G W + keep .* (Q A)).  A template parameter, not a runtime test: the
// keep-mask hashes depend only on (m, n), so the compiler hoists all 64 of them (+ their SGPR-pair results, spilled to
// VGPR lanes) to the top of the kernel -- ~700 instructions per workgroup that the forward / P / Q launches never use.
// NW = waves per workgroup.  The 128 x 128 tile is unchanged; with 8 waves a wave owns 64 n x 32 m (half the
// accumulators, half the staging registers, half the loads / LDS traffic / MFMAs per step), fits 128 VGPRs and runs
// 4 waves per SIMD instead of 2 -- the kernel is latency- and issue-bound, not bandwidth-bound.
// d/dh of the exact (erf) GELU, the factor ATen's GeluBackward applies: Phi(h) + h * phi(h)
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) sharing its exp(-h^2/2) with the density:
// ~16 VALU operations per element -- with ocml's erff + expf (~60) the epilogue of the hidden-width dX launches became
// VALU-bound and gave back most of the saved pass.
__device__ __forceinline__ float gelu_grad(float h) {
    const float z = fabsf(h) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
    const float e = __expf(-z * z);  // exp(-h^2 / 2)
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float erf_abs = 1.f - p * t * e;  // erf(|h| / sqrt 2)
    const float cdf = 0.5f + 0.5f * copysignf(erf_abs, h);
    return cdf + h * e * 0.39894228040143268f;
}

// exact (erf) GELU with the same erf: h * Phi(h)
__device__ __forceinline__ float gelu_fwd(float h) {
    const float z = fabsf(h) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float erf_abs = 1.f - p * t * __expf(-z * z);
    return h * (0.5f + 0.5f * copysignf(erf_abs, h));
}

template <typename T, bool MULTI, bool MS, bool MLR, int NW, bool GATE = false, bool ACT = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void k_nt(const NtParams Pv) {
    constexpr int SM = 8 / NW;       // 32-row m sub-blocks per wave
    constexpr int MW = 32 * SM;      // m rows per wave
    constexpr int NT = 64 * NW;      // threads
    (void)Pv;
    NtPtr P = (NtPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int KE = ROWB / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // staging (2 x 128 x LDSB)
    unsigned char* sW = smem;
    unsigned char* sA = smem + TILE * LDSB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = NW == 4 ? wave >> 1 : wave >> 2, wm = NW == 4 ? wave & 1 : wave & 3;

    // batched form
    const void* act0 = P->act[0];
    int row_off = 0, n_rows = P->n_rows;
    bool act_mask = P->act_mask != 0;
    if (P->nz > 0) {
        const int z = blockIdx.z;
        act0 = P->zact[z];
        row_off = P->zrow0[z];
        n_rows = P->zrows[z];
        act_mask = P->zmask[z] != 0;
    }
    if (n_rows <= 0) return;
    if (P->dbg & NT_DBG_MASK & 16) return;
    act_mask = act_mask && P->drop.thr16 != 0;

    // XCD-aware tile order: hardware places block b on XCD b % 8; give every XCD a contiguous run of
    // logical tiles so that the n-tiles sharing one activation row-block hit the same L2 (T1, bijective).
    const int n_tiles = (n_rows + TILE - 1) / TILE;
    const int64_t m_tiles = (P->M + TILE - 1) / TILE;
    const int64_t nwg = m_tiles * n_tiles;
    int64_t b = blockIdx.x;
    if (b >= nwg) return;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = b % 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const int64_t bm = b / n_tiles;
    const int bn0 = (int)(b % n_tiles);
    const int64_t m0 = bm * TILE;
    const int n0 = bn0 * TILE;

    const T* wgt = reinterpret_cast<const T*>(P->wgt) + (int64_t)row_off * P->ld_wgt;
    DropoutCfg drop;
    drop.seed_lo = P->drop.seed_lo;
    drop.seed_hi = P->drop.seed_hi;
    drop.thr16 = P->drop.thr16;
    drop.off = P->drop.off;
    mtl_dropout_resolve(drop);
    const int nseq = MULTI ? 1 + P->n_out : 2 * P->n_out;

    // ---- loader side of the stream (register prefetch, one wide tile ahead)
    TileRegs<T, SM> rg;
    NtCursor ld = nt_seek<MULTI>(P, 0, nseq, bn0);
    auto issue = [&](const NtCursor& c) __attribute__((always_inline)) {
        if (c.lr)  // rank segment: weights = Rm, activation = L
            nt_load<T, false, SM>(rg, tid, reinterpret_cast<const T*>(P->Rm), P->ldR, c.bn * TILE, n_rows, P->L, P, 1, P->ldL, m0, P->M, c.k0,
                                  c.k_hi, false);
        else
            nt_load<T, MS, SM>(rg, tid, wgt, P->ld_wgt, c.bn * TILE, n_rows, act0, P, P->n_act, P->ld_act, m0, P->M, c.k0, c.k_hi, act_mask);
    };
    const int dbg = P->dbg & NT_DBG_MASK;
    if (ld.valid && !(dbg & 2)) issue(ld);
    // consume one tile: registers -> LDS, prefetch the next tile of the stream, multiply
    auto step = [&](f32x16(&acc)[2][SM], int k_left) __attribute__((always_inline)) {
        if (!(dbg & 64)) nt_store_lds<T, SM>(rg, tid, sW, sA, drop, m0);
        if (!(dbg & 128)) __syncthreads();
        ld.k0 += KE;
        if (ld.k0 >= ld.k_hi) ld = nt_seek<MULTI>(P, ld.q + 1, nseq, bn0);
        if (ld.valid && !(dbg & 2)) issue(ld);
        // a wave whose 64 output columns lie entirely past n_rows (P / Q passes: <= 64 of the tile's 128 columns exist)
        // only helps staging: no LDS fragment reads, no MFMAs (wave-uniform test)
        if (n0 + wn * 64 < n_rows && !(dbg & 4)) nt_compute<T, SM>(acc, sW, sA, lane, wn, wm, k_left);
        if (!(dbg & 128)) __syncthreads();
    };
    auto run_part = [&](int q, f32x16(&acc)[2][SM]) __attribute__((always_inline)) {
        int lr, lo, hi;
        nt_part<MULTI>(P, q, lr, lo, hi);
        for (int k0 = lo; k0 < hi; k0 += KE) step(acc, hi - k0);
        return hi > lo;
    };
    auto zero = [](f32x16(&a)[2][SM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < SM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[i][j][r] = 0.f;
    };
    // acc = acc * alpha[n] + bias[n]
    auto affine = [&](f32x16(&a)[2][SM]) __attribute__((always_inline)) {
        if (!(P->alpha || P->bias) || (P->dbg & NT_DBG_MASK & 32)) return;
#pragma unroll
        for (int sn = 0; sn < 2; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * (lane >> 5);
                if (n < n_rows) {
                    f32x4 al = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                    if (P->alpha) al = *reinterpret_cast<const f32x4*>(P->alpha + row_off + n);
                    if (P->bias) bi = *reinterpret_cast<const f32x4*>(P->bias + row_off + n);
#pragma unroll
                    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[sn][sm][q * 4 + e] = a[sn][sm][q * 4 + e] * al[e] + bi[e];
                }
            }
    };
    // acc *= keep(m, n)
    auto apply_mask = [&](f32x16(&a)[2][SM]) __attribute__((always_inline)) {
#pragma unroll
        for (int sm = 0; sm < SM; ++sm) {
            const int64_t m = m0 + wm * MW + sm * 32 + (lane & 31);
            const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)m);
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * (lane >> 5);
                    const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                    const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                    if ((h0 & 0xFFFFu) < drop.thr16) a[sn][sm][q * 4 + 0] = 0.f;
                    if ((h0 >> 16) < drop.thr16) a[sn][sm][q * 4 + 1] = 0.f;
                    if ((h1 & 0xFFFFu) < drop.thr16) a[sn][sm][q * 4 + 2] = 0.f;
                    if ((h1 >> 16) < drop.thr16) a[sn][sm][q * 4 + 3] = 0.f;
                }
        }
    };
    auto store = [&](const f32x16(&a)[2][SM], void* ptr, const void* gate_ptr, void* act_ptr) __attribute__((always_inline)) {
        T* outp = reinterpret_cast<T*>(ptr);
        const T* gate = reinterpret_cast<const T*>(gate_ptr);
        T* actp = reinterpret_cast<T*>(act_ptr);
        (void)gate;
        (void)actp;
        if (!outp || n0 + wn * 64 >= n_rows || (dbg & 8)) return;  // (the per-wave LDS image needs no workgroup barrier)
        if constexpr (sizeof(T) == 2) {
            // bf16: transpose the wave's 64(n) x 64(m) accumulator tile through LDS so that every store instruction
            // writes whole 128-byte row segments (8 lanes x 16 B) instead of 16-byte pieces of 32 different rows.
            // The staging buffers are idle here (the trailing barrier of the last tile has passed).
            constexpr int ORS = EPI_ROW;  // row stride of the per-wave image (bytes): 2-way conflicts at most
            unsigned char* img = smem + wave * (MW * ORS);
#pragma unroll
            for (int sm = 0; sm < SM; ++sm)
#pragma unroll
                for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ml = sm * 32 + (lane & 31), nl = sn * 32 + 8 * q + 4 * (lane >> 5);
                        u32x2 pk = {mtl_pack2<T>(a[sn][sm][q * 4], a[sn][sm][q * 4 + 1]),
                                    mtl_pack2<T>(a[sn][sm][q * 4 + 2], a[sn][sm][q * 4 + 3])};
                        *reinterpret_cast<u32x2*>(img + ml * ORS + nl * 2) = pk;
                    }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is private to this wave, no barrier needed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4 * SM; ++it) {
                const int ml = it * 8 + (lane >> 3), c16 = lane & 7;
                const int64_t m = m0 + wm * MW + ml;
                const int n = n0 + wn * 64 + c16 * 8;
                u32x4 v = *reinterpret_cast<const u32x4*>(img + ml * ORS + c16 * 16);
                if (m < P->M && n < n_rows && !(dbg & 1)) {
                    if constexpr (GATE) {
                        if (gate) {  // the bf16-rounded gradient times gelu'(pre-activation), rounded once (as ATen does)
                            const u32x4 hv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gate + m * P->ld_out + row_off + n));
                            v = mtl_gelu_gate_pk4<T, false>(v, hv);
                        }
                    }
                    // non-temporal: the 77 - 308 MB outputs of a launch outlive L2 / MALL anyway (+1 % on the step; the same hint
                    // on the glue kernels' stores costs 1.5 %: their consumers do hit in cache)
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(outp + m * P->ld_out + row_off + n));
                    if constexpr (ACT) {
                        if (actp) {  // second output: GELU of the bf16-rounded value, rounded once (ATen's gelu on the bf16 tensor)
                            const u32x4 av = mtl_gelu_pk4<T, false>(v);
                            __builtin_nontemporal_store(av, reinterpret_cast<u32x4*>(actp + m * P->ld_out + row_off + n));
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int sm = 0; sm < SM; ++sm) {
                const int64_t m = m0 + wm * MW + sm * 32 + (lane & 31);
#pragma unroll
                for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wn * 64 + sn * 32 + 8 * q + 4 * (lane >> 5);
                        if (m < P->M && n < n_rows) {
                            T* dst = outp + m * P->ld_out + row_off + n;
                            f32x4 o4 = {a[sn][sm][q * 4], a[sn][sm][q * 4 + 1], a[sn][sm][q * 4 + 2], a[sn][sm][q * 4 + 3]};
                            if constexpr (GATE) {
                                if (gate) {
                                    const f32x4 hv = *reinterpret_cast<const f32x4*>(gate + m * P->ld_out + row_off + n);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o4[e] *= gelu_grad(hv[e]);
                                }
                            }
                            *reinterpret_cast<f32x4*>(dst) = o4;
                            if constexpr (ACT) {
                                if (actp) {
                                    f32x4 a4;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) a4[e] = gelu_fwd(o4[e]);
                                    *reinterpret_cast<f32x4*>(actp + m * P->ld_out + row_off + n) = a4;
                                }
                            }
                        }
                    }
            }
        }
    };

    if constexpr (MULTI) {
        f32x16 base[2][SM], acc[2][SM];
        zero(base);
        run_part(0, base);
        affine(base);
        for (int o = 0; o < P->n_out; ++o) {
            const NtOut O = nt_out(P, o);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < SM; ++j) acc[i][j] = base[i][j];
            run_part(1 + o, acc);
            if (O.fold) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < SM; ++j) base[i][j] = acc[i][j];
            }
            store(acc, O.ptr, O.gate, O.act);
            __syncthreads();  // the output image lives in the staging buffers
        }
    } else {
        f32x16 acc[2][SM];
        for (int o = 0; o < P->n_out; ++o) {
            const NtOut O = nt_out(P, o);
            zero(acc);
            const bool had_lr = run_part(2 * o, acc);
            if constexpr (MLR) {
                if (had_lr && O.mask_lr && drop.enabled()) apply_mask(acc);
            } else {
                (void)had_lr;
            }
            run_part(2 * o + 1, acc);
            if (O.use_base) affine(acc);
            store(acc, O.ptr, O.gate, O.act);
            __syncthreads();  // the output image lives in the staging buffers
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_ntl : the lean bf16 launches of k_nt (ONE output, one activation source, no batched / row-panel form: every P / Q pass,
// every T = 0 forward and dX) as straight-line code.
// Why a second kernel: an ablation of k_nt (MTLORA_NT_DBG, tools/nt_ablate.sh; profiles/r02_nt_ablate.txt) showed that with
// loads, MFMAs, stores and the epilogue all switched OFF the s0.qkv forward still took 100 of its 164 us -- the generic
// kernel's bookkeeping.  Its prologue is a chain of ~15 DEPENDENT scalar loads from the 1 KB parameter block (each followed by
// s_waitcnt lgkmcnt(0)), three software 64-bit divisions (~130 SALU instructions each) for the XCD map, a part-sequence cursor
// that re-reads the output table per step, and every global load sits in its own exec-masked branch.  With only 2 - 4 k-steps
// per workgroup nothing amortises that.  Here:
//   * a compact parameter block (~200 B) read once; the XCD map and tile decomposition use 32-bit arithmetic and a host-side
//     magic multiplier instead of divisions;
//   * the step sequence is [rank-segment k-tiles | base k-tiles], both counts known up front: no cursor;
//   * loads are unconditional: out-of-range rows are CLAMPED (their products land in rows / columns that are never stored)
//     and out-of-range k vectors read a 16-byte zero page -- address selects, no branches;
//   * the bias is the accumulator's initial value (its loads overlap the first tile's) instead of a dependent load + FMA pass
//     after the last MFMA.
// Tile geometry, LDS layout, fragment reads and the transposing epilogue are k_nt's 8-wave variant (128 x 128 x 96, wave tile
// 64 n x 32 m), so results are bit-identical to k_nt's.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

struct NlParams {
    const bf16* act;    // (M x K) activation-like operand of the base part
    const bf16* wgt;    // (n_rows x K)
    const bf16* L;      // (M x ldL) activation-like operand of the rank part
    const bf16* Rm;     // (n_rows x ldR)
    bf16* out;          // (M x ld_out)
    bf16* act2;         // ACT: second output gelu(out)
    const bf16* gate;   // GATE (k_ntd): out *= gelu'(gate[m][n]), same layout as out
    const float* bias;  // per output column, nullable
    const float* alpha; // per output column multiplier, nullable
    int64_t ld_act, ld_wgt, ldL, ldR, ld_out;
    int M, n_rows, K, seg_lo, seg_hi;
    int n_tiles;
    uint32_t nt_magic;  // floor(2^32 / n_tiles) + 1: b / n_tiles == umulhi(b, nt_magic) for b * n_tiles < 2^32 (n_tiles > 1)
    uint32_t q8, r8;    // workgroups / 8, workgroups % 8 (XCD map)
    int act_mask, use_base;
    int dbg, pad_;
    DropoutCfg drop;
};

// SN = 32-column sub-blocks per wave: 2 (128 x 128 tile) or 3 (128 rows x 192 columns).  The wide tile exists for the launches
// whose 128 x 128 tile count lands just above a whole number of residency rounds (2 workgroups per CU = 512 slots): the
// N = 384 outputs of stage 2 (196 x 3 = 588 tiles = 1.15 rounds -> 2 rounds, the second one 15 % full) run as 196 x 2 = 392
// tiles of 1.5x the work in ONE round.  MLR: the low-rank part (rank tiles come first in the stream) is multiplied by the dropout
// keep-mask of (m, n) before the base tiles are added -- the dX launches (dX = keep .* (Q A) + dY W).
template <bool ACT, bool MLR, int SN>
__global__ __launch_bounds__(512, 4) void k_ntl(const NlParams P) {
    constexpr int KE = ROWB / 2;  // 96 elements per staged k-tile
    constexpr int TN = 64 * SN;   // tile columns
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;
    unsigned char* sA = smem + TN * LDSB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wm = wave & 3;

    // XCD-aware tile order (as k_nt): block b runs on XCD b % 8; every XCD gets a contiguous run of logical tiles
    uint32_t b = blockIdx.x;
    {
        const uint32_t xcd = b & 7u, q = P.q8, r = P.r8;
        b = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (b >> 3);
    }
    const uint32_t bm = P.n_tiles == 1 ? b : __umulhi(b, P.nt_magic);
    const int bn = (int)(b - bm * (uint32_t)P.n_tiles);
    const int m0 = (int)bm * TILE, n0 = bn * TN;
    const int M = P.M, n_rows = P.n_rows;

    DropoutCfg drop = P.drop;
    mtl_dropout_resolve(drop);
    const bool act_mask = P.act_mask != 0 && drop.thr16 != 0;
    const int dbg = P.dbg & NT_DBG_MASK;

    const int n1 = P.seg_hi > P.seg_lo ? (P.seg_hi - P.seg_lo + KE - 1) / KE : 0;
    const int n2 = (P.use_base && P.K > 0) ? (P.K + KE - 1) / KE : 0;
    const int total = n1 + n2;

    f32x16 acc[SN];
    // accumulator start: the bias (when nothing multiplies the sum afterwards and no mask is applied to the running sum)
    const bool bias_first = !MLR && P.bias != nullptr && P.alpha == nullptr && P.use_base != 0;
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 bi = {0.f, 0.f, 0.f, 0.f};
            if (bias_first) {
                int n = n0 + wn * (32 * SN) + sn * 32 + 8 * q + 4 * (lane >> 5);
                n = n < n_rows - 4 ? n : n_rows - 4;  // (columns >= n_rows are never stored)
                bi = *reinterpret_cast<const f32x4*>(P.bias + n);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[sn][q * 4 + e] = bi[e];
        }

    u32x4 rw[3], rw2[SN == 3 ? 3 : 1], ra[3];  // staged k-tile: weight rows 0..127, weight rows 128..191 (SN = 3), activation rows
    (void)rw2;
    int cur_mask = 0, cur_k0 = 0;  // of the tile sitting in the registers
    auto issue = [&](int i) __attribute__((always_inline)) {
        const bool lr = i < n1;
        const bf16* wp = lr ? P.Rm : P.wgt;
        const bf16* ap = lr ? P.L : P.act;
        const int64_t ldw = lr ? P.ldR : P.ld_wgt, lda = lr ? P.ldL : P.ld_act;
        const int k0 = lr ? P.seg_lo + i * KE : (i - n1) * KE;
        const int khi = lr ? P.seg_hi : P.K;
        cur_mask = (!lr && act_mask) ? 1 : 0;
        cur_k0 = k0;
        const bf16* zp = reinterpret_cast<const bf16*>(g_zero16);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            int r, v;
            nt_map<1>(tid, sl, r, v);
            const int k = k0 + v * 8;
            const bool kin = k < khi;
            int wr = n0 + r, ar = m0 + r;
            wr = wr < n_rows ? wr : n_rows - 1;
            ar = ar < M ? ar : M - 1;
            rw[sl] = *reinterpret_cast<const u32x4*>(kin ? wp + (int64_t)wr * ldw + k : zp);
            ra[sl] = *reinterpret_cast<const u32x4*>(kin ? ap + (int64_t)ar * lda + k : zp);
            if constexpr (SN == 3) {  // weight rows 128 .. 191: the same map on a second 128-row panel, upper half unused
                int wr2 = n0 + 128 + r;
                wr2 = wr2 < n_rows ? wr2 : n_rows - 1;
                if (r < 64) rw2[sl] = *reinterpret_cast<const u32x4*>(kin ? wp + (int64_t)wr2 * ldw + k : zp);
            }
        }
    };
    auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            int r, v;
            nt_map<1>(tid, sl, r, v);
            if (cur_mask) {  // uniform
                const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(m0 + r));
                VOps<bf16>::drop(ra[sl], drop, rh, (uint32_t)(cur_k0 + v * 8));
            }
            *reinterpret_cast<u32x4*>(sW + r * LDSB + v * 16) = rw[sl];
            *reinterpret_cast<u32x4*>(sA + r * LDSB + v * 16) = ra[sl];
            if constexpr (SN == 3) {
                if (r < 64) *reinterpret_cast<u32x4*>(sW + (128 + r) * LDSB + v * 16) = rw2[sl];
            }
        }
    };
    auto compute = [&](int k_left) __attribute__((always_inline)) {
        const int h = lane >> 5, rl = lane & 31;
#pragma unroll
        for (int t = 0; t < SUBT; ++t) {
            if (t * 32 >= k_left) break;
            Frag<bf16> fw[SN], fa;
#pragma unroll
            for (int sn = 0; sn < SN; ++sn) {
                const unsigned char* pw = sW + (wn * (32 * SN) + sn * 32 + rl) * LDSB + t * 64;
                fw[sn].v[0] = *reinterpret_cast<const u32x4*>(pw + h * 16);
                fw[sn].v[1] = *reinterpret_cast<const u32x4*>(pw + (2 + h) * 16);
            }
            const unsigned char* pa = sA + (wm * 32 + rl) * LDSB + t * 64;
            fa.v[0] = *reinterpret_cast<const u32x4*>(pa + h * 16);
            fa.v[1] = *reinterpret_cast<const u32x4*>(pa + (2 + h) * 16);
#pragma unroll
            for (int sn = 0; sn < SN; ++sn) mtl_mma(fw[sn], fa, acc[sn]);
        }
    };
    const bool wave_live = n0 + wn * (32 * SN) < n_rows;  // P / Q passes: a wave whose columns do not exist only helps staging
    if (total > 0 && !(dbg & 2)) issue(0);
    for (int i = 0; i < total; ++i) {
        const bool lr = i < n1;
        const int k_left = lr ? P.seg_hi - (P.seg_lo + i * KE) : P.K - (i - n1) * KE;
        if (!(dbg & 64)) stage();
        __syncthreads();
        if (i + 1 < total && !(dbg & 2)) issue(i + 1);
        if (wave_live && !(dbg & 4)) compute(k_left);
        if constexpr (MLR) {
            if (i + 1 == n1 && wave_live && drop.thr16 != 0) {  // the rank part is complete: acc *= keep(m, n)
                const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(m0 + wm * 32 + (lane & 31)));
#pragma unroll
                for (int sn = 0; sn < SN; ++sn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wn * (32 * SN) + sn * 32 + 8 * q + 4 * (lane >> 5);
                        const uint32_t h0 = mtl_dropout_pairbits(drop, rh, (uint32_t)n);
                        const uint32_t h1 = mtl_dropout_pairbits(drop, rh, (uint32_t)(n + 2));
                        if ((h0 & 0xFFFFu) < drop.thr16) acc[sn][q * 4 + 0] = 0.f;
                        if ((h0 >> 16) < drop.thr16) acc[sn][q * 4 + 1] = 0.f;
                        if ((h1 & 0xFFFFu) < drop.thr16) acc[sn][q * 4 + 2] = 0.f;
                        if ((h1 >> 16) < drop.thr16) acc[sn][q * 4 + 3] = 0.f;
                    }
            }
        }
        __syncthreads();
    }
    if (!wave_live || (dbg & 8)) return;

    if (!bias_first && (P.alpha || P.bias) && P.use_base) {  // acc = acc * alpha[n] + bias[n]
#pragma unroll
        for (int sn = 0; sn < SN; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int n = n0 + wn * (32 * SN) + sn * 32 + 8 * q + 4 * (lane >> 5);
                n = n < n_rows - 4 ? n : n_rows - 4;
                f32x4 al = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                if (P.alpha) al = *reinterpret_cast<const f32x4*>(P.alpha + n);
                if (P.bias) bi = *reinterpret_cast<const f32x4*>(P.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[sn][q * 4 + e] = acc[sn][q * 4 + e] * al[e] + bi[e];
            }
    }

    // epilogue: transpose the wave's (32 SN) (n) x 32 (m) tile through a private LDS image -> whole row-segment stores
    {
        constexpr int ORS = 64 * SN + 8;       // image row stride (bytes)
        constexpr int CPRW = 4 * SN;           // 16-byte chunks per image row
        unsigned char* img = smem + wave * (32 * ORS);
#pragma unroll
        for (int sn = 0; sn < SN; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ml = lane & 31, nl = sn * 32 + 8 * q + 4 * (lane >> 5);
                u32x2 pk = {mtl_pack_bf16(acc[sn][q * 4], acc[sn][q * 4 + 1]), mtl_pack_bf16(acc[sn][q * 4 + 2], acc[sn][q * 4 + 3])};
                *reinterpret_cast<u32x2*>(img + ml * ORS + nl * 2) = pk;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is private to this wave
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2 * SN; ++it) {
            const int idx = it * 64 + lane;
            const int ml = idx / CPRW, c16 = idx - ml * CPRW;
            const int m = m0 + wm * 32 + ml;
            const int n = n0 + wn * (32 * SN) + c16 * 8;
            u32x4 v = *reinterpret_cast<const u32x4*>(img + ml * ORS + c16 * 16);
            if (m < M && n < n_rows && !(dbg & 1)) {
                const int64_t o = (int64_t)m * P.ld_out + n;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(P.out + o));
                if constexpr (ACT) {
                    if (P.act2) {
                        const u32x4 av = mtl_gelu_pk4<bf16, false>(v);
                        __builtin_nontemporal_store(av, reinterpret_cast<u32x4*>(P.act2 + o));
                    }
                }
            }
        }
    }
}

#include "stream.h"
#include "dense.h"
#include "pq.h"

// ------------------------------------------------------------------------------------------------
// k_tn : Out[a][b] = sum_m SrcA[m][a0 + a] * SrcB[m][b0 + b], split over m.
// SrcA is always the NARROW (rank-side) operand (Q or P, <= 64 columns per tile) and SrcB the WIDE one
// (X or dY, 256 columns per tile), so a workgroup streams 64 + 256 columns per row and the wide matrix is
// read ~once (x1.25 with the narrow slab) instead of twice with square tiles.  dB = dY^T P is computed as
// its transpose P^T dY and written back transposed by k_tn_reduce.
// ------------------------------------------------------------------------------------------------
constexpr int TN_A = 64;
constexpr int TN_B = 256;
constexpr int TN_TILE = TN_A * TN_B;
struct TnProblem {
    const void* A;
    const void* B;
    int64_t lda, ldb;
    int a0, Na, b0, Nb;  // column windows
    int b_mask;          // dropout keep-mask on SrcB (keyed by (m, b0 + b))
    int tiles_a, tiles_b;
    float* part;         // [nsplit][tiles_a*tiles_b][64*256]
    float* out;          // fp32; element (a, b) at out[a*ldo + b], or out[b*ldo + a] when transpose
    int out_a, out_b, ldo, transpose;
};
struct TnParams {
    TnProblem p[2 * MAXO];
    int n_prob;
    int64_t M;
    int nsplit;
    int64_t rows_per_split;
    DropoutCfg drop;
};

template <typename T>
struct TnCfg;
template <>
struct TnCfg<bf16> {
    static constexpr int SUB = 32;  // rows (m) per MFMA k-tile
};
template <>
struct TnCfg<f16> {
    static constexpr int SUB = 32;
};
template <>
struct TnCfg<float> {
    static constexpr int SUB = 16;
};

// transposed fragment: lane (i = l & 31, h = l >> 5) gets Src[m = slot(h, e)][col0 + i]; ``lr`` = LDS row bytes
template <typename H>  // any 16-bit element type (the transposing read moves bits)
__device__ __forceinline__ Frag<H> tn_frag16(const unsigned char* s, int col0, int lane, int lr) {
    // ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the 8-byte address of row (i>>2),
    // columns 4*(i&3)..+3 of a [4][16] block and receives column i of that block (4 rows).
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int col = col0 + 16 * (g & 1) + 4 * (i & 3);
    Frag<H> f;
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // rows: {8h+0..3}, {8h+4..7}, {16+8h+0..3}, {16+8h+4..7}
        const int row = ((j >> 1) * 16) + 8 * h + 4 * (j & 1) + (i >> 2);
        const unsigned char* p = s + row * lr + col * 2;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(p));
        u32x2 u = __builtin_bit_cast(u32x2, v);
        w[2 * j] = u[0];
        w[2 * j + 1] = u[1];
    }
    f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
    f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
    return f;
}
__device__ __forceinline__ Frag<bf16> tn_frag(const unsigned char* s, int col0, int lane, int lr, bf16*) {
    return tn_frag16<bf16>(s, col0, lane, lr);
}
__device__ __forceinline__ Frag<f16> tn_frag(const unsigned char* s, int col0, int lane, int lr, f16*) {
    return tn_frag16<f16>(s, col0, lane, lr);
}
__device__ __forceinline__ Frag<float> tn_frag(const unsigned char* s, int col0, int lane, int lr, float*) {
    const int h = lane >> 5, i = lane & 31;
    Frag<float> f;
    uint32_t w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {  // rows {4h..4h+3} U {8+4h..8+4h+3}
        const int row = (e >> 2) * 8 + 4 * h + (e & 3);
        w[e] = *reinterpret_cast<const uint32_t*>(s + row * lr + (col0 + i) * 4);
    }
    f.v[0] = u32x4{w[0], w[1], w[2], w[3]};
    f.v[1] = u32x4{w[4], w[5], w[6], w[7]};
    return f;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void k_tn(const TnParams P) {
    constexpr int SUB = TnCfg<T>::SUB;
    constexpr int KE = 2 * SUB;  // rows per staged chunk
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = ET<T>::VEC;
    // padded LDS rows.  bf16: a ds_read_b64_tr_b16 cycle serves 32 lanes = 4 rows x 2 column halves of 32 B; the 8
    // segments fall in distinct bank groups iff the row stride is 16 dwords (mod 64): 320 B / 576 B (strides of
    // 144 B / 528 B cost 42 % of the LDS cycles in conflicts).  f32 (scalar 4-byte reads across 32 columns): +16 B.
    constexpr int LRA = ES == 2 ? 320 : TN_A * ES + 16, LRB = ES == 2 ? 576 : TN_B * ES + 16;
    constexpr int VPA = TN_A / VEC, VPB = TN_B / VEC;          // 16-byte vectors per tile row
    constexpr int RSA = 256 / VPA, RSB = 256 / VPB;            // rows covered by one sweep of the workgroup
    constexpr int NLA = KE / RSA, NLB = KE / RSB;              // loads per thread per chunk (2 and 8)
    __shared__ __attribute__((aligned(16))) unsigned char smem[KE * (LRA + LRB)];
    unsigned char* sA = smem;
    unsigned char* sB = smem + KE * LRA;
    const TnProblem& pr = P.p[blockIdx.z];
    const int tile = blockIdx.y;
    if (tile >= pr.tiles_a * pr.tiles_b) return;
    const int ta = tile / pr.tiles_b, tb = tile % pr.tiles_b;
    const int split = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const int64_t m_lo = (int64_t)split * P.rows_per_split;
    int64_t m_hi = m_lo + P.rows_per_split;
    if (m_hi > P.M) m_hi = P.M;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int rowA = tid / VPA, vecA = tid % VPA, rowB = tid / VPB, vecB = tid % VPB;
    const int ca = ta * TN_A + vecA * VEC;  // column inside the A window
    const int cb = tb * TN_B + vecB * VEC;
    const bool a_in = ca < pr.Na, b_in = cb < pr.Nb;
    const bool wave_on = tb * TN_B + wave * 64 < pr.Nb;  // this wave's 64 wide columns hold data
    const T* Ap = reinterpret_cast<const T*>(pr.A) + pr.a0 + ca;
    const T* Bp = reinterpret_cast<const T*>(pr.B) + pr.b0 + cb;
    DropoutCfg drop = P.drop;
    mtl_dropout_resolve(drop);
    const bool bmask = pr.b_mask && drop.enabled();

    // two register sets = prefetch distance 2 chunks (the grid is sized to 2 workgroups per CU, i.e. 256 VGPRs per
    // wave, and a workgroup's streaming rate is bounded by bytes in flight / load latency)
    u32x4 ra0[NLA], rb0[NLB], ra1[NLA], rb1[NLB];
    auto load = [&](u32x4* ra, u32x4* rb, int64_t mrow) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLA; ++j) {
            const int64_t m = mrow + rowA + j * RSA;
            ra[j] = (a_in && m < m_hi) ? *reinterpret_cast<const u32x4*>(Ap + m * pr.lda) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < NLB; ++j) {
            const int64_t m = mrow + rowB + j * RSB;
            rb[j] = (b_in && m < m_hi) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Bp + m * pr.ldb)) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto stage = [&](u32x4* ra, u32x4* rb, int64_t mrow) __attribute__((always_inline)) {
        if (bmask && b_in) {  // dropout keep-mask, applied just before the LDS store
#pragma unroll
            for (int j = 0; j < NLB; ++j) {
                const uint32_t rh = mtl_dropout_rowhash(drop, 0u, (uint32_t)(mrow + rowB + j * RSB));
                Vec16<T> x;
                x.raw = rb[j];
#pragma unroll
                for (int e = 0; e < VEC; e += 2) {
                    const uint32_t h = mtl_dropout_pairbits(drop, rh, (uint32_t)(pr.b0 + cb + e));
                    if ((h & 0xFFFFu) < drop.thr16) x.e[e] = mtl_from_f32<T>(0.f);
                    if ((h >> 16) < drop.thr16) x.e[e + 1] = mtl_from_f32<T>(0.f);
                }
                rb[j] = x.raw;
            }
        }
#pragma unroll
        for (int j = 0; j < NLA; ++j) *reinterpret_cast<u32x4*>(sA + (rowA + j * RSA) * LRA + vecA * 16) = ra[j];
#pragma unroll
        for (int j = 0; j < NLB; ++j) *reinterpret_cast<u32x4*>(sB + (rowB + j * RSB) * LRB + vecB * 16) = rb[j];
    };
    auto compute = [&]() __attribute__((always_inline)) {
        if (!wave_on) return;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const unsigned char* a_s = sA + sub * SUB * LRA;
            const unsigned char* b_s = sB + sub * SUB * LRB;
            Frag<T> fa0 = tn_frag(a_s, 0, lane, LRA, (T*)nullptr);
            Frag<T> fa1 = tn_frag(a_s, 32, lane, LRA, (T*)nullptr);
            Frag<T> fb0 = tn_frag(b_s, wave * 64, lane, LRB, (T*)nullptr);
            Frag<T> fb1 = tn_frag(b_s, wave * 64 + 32, lane, LRB, (T*)nullptr);
            mtl_mma(fa0, fb0, acc[0][0]);
            mtl_mma(fa0, fb1, acc[0][1]);
            mtl_mma(fa1, fb0, acc[1][0]);
            mtl_mma(fa1, fb1, acc[1][1]);
        }
    };

    if (m_lo < m_hi) load(ra0, rb0, m_lo);
    if (m_lo + KE < m_hi) load(ra1, rb1, m_lo + KE);
    for (int64_t mrow = m_lo; mrow < m_hi; mrow += 2 * KE) {
        stage(ra0, rb0, mrow);
        __syncthreads();
        if (mrow + 2 * KE < m_hi) load(ra0, rb0, mrow + 2 * KE);
        compute();
        __syncthreads();
        if (mrow + KE < m_hi) {
            stage(ra1, rb1, mrow + KE);
            __syncthreads();
            if (mrow + 3 * KE < m_hi) load(ra1, rb1, mrow + 3 * KE);
            compute();
            __syncthreads();
        }
    }

    if (!wave_on) return;  // k_tn_reduce never reads columns outside the B window
    float* dst = pr.part + ((int64_t)split * (pr.tiles_a * pr.tiles_b) + tile) * TN_TILE;
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ia * 32 + mtl_d_row(lane, r), j = wave * 64 + jb * 32 + mtl_d_col(lane);
                dst[i * TN_B + j] = acc[ia][jb][r];
            }
}

constexpr int TN_RG = 4;  // thread groups that share the splits of a 1024-element block
__global__ __launch_bounds__(256 * TN_RG) void k_tn_reduce(const TnParams P) {
    // one workgroup per (problem, tile, 1024-element block): 4 outputs per thread (one 16-byte load per split, coalesced
    // across the wave); the splits are dealt round-robin to TN_RG groups of 256 threads, each summing its share in a
    // fixed order with 4 independent chains, then a fixed-order LDS combine (deterministic)
    __shared__ f32x4 sm[TN_RG][256];
    const TnProblem& pr = P.p[blockIdx.z];
    const int ntile = pr.tiles_a * pr.tiles_b;
    const int tile = blockIdx.y;
    if (tile >= ntile) return;
    const int t = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int e0 = blockIdx.x * 1024 + t * 4;  // element inside the 64x256 tile
    const int ta = tile / pr.tiles_b, tb = tile % pr.tiles_b;
    const int a = ta * TN_A + e0 / TN_B, b0 = tb * TN_B + e0 % TN_B;
    const bool live = a < pr.out_a && b0 < pr.out_b;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* src = pr.part + (int64_t)tile * TN_TILE + e0;
        const int64_t stride = (int64_t)ntile * TN_TILE;
        int sp = grp;
        for (; sp + 3 * TN_RG < P.nsplit; sp += 4 * TN_RG) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const f32x4*>(src + (int64_t)(sp + u * TN_RG) * stride);
        }
        for (; sp < P.nsplit; sp += TN_RG) acc[0] += *reinterpret_cast<const f32x4*>(src + (int64_t)sp * stride);
    }
    sm[grp][t] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (grp != 0 || !live) return;
    f32x4 tsum = sm[0][t];
#pragma unroll
    for (int gI = 1; gI < TN_RG; ++gI) tsum += sm[gI][t];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (b0 + e < pr.out_b) {
            const int64_t at = pr.transpose ? (int64_t)(b0 + e) * pr.ldo + a : (int64_t)a * pr.ldo + b0 + e;
            pr.out[at] = tsum[e];
        }
}

// elementwise sum of up to MAXO tensors (matrixv2 backward: G for the shared factors)
struct SumParams {
    const void* src[MAXO];
    int n;
    int64_t nvec;
};
template <typename T>
__global__ __launch_bounds__(256) void k_sum(SumParams P, T* out) {
    constexpr int VEC = ET<T>::VEC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.nvec; i += (int64_t)gridDim.x * 256) {
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = 0.f;
        for (int s = 0; s < P.n; ++s) {
            Vec16<T> y = mtl_ld16<T>(reinterpret_cast<const T*>(P.src[s]) + i * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] += mtl_to_f32(y.e[e]);
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.e[e] = mtl_from_f32<T>(f[e]);
        *reinterpret_cast<u32x4*>(out + i * VEC) = o.raw;
    }
}

// ------------------------------------------------------------------------------------------------
// k_rank_out : the task outputs of a dX launch,  dX_t = Q_t A_t  [.* gelu'(h_t)],  for SMALL task ranks (rp <= 16).
// They share nothing with the base GEMM (no dY W term), and with r_t = 4 the "GEMM" is 8 multiply-adds per element: in the tiled
// multi-output kernel each of them costs a full tile pass (rank k-tile staging, MFMA on a mostly-zero k-tile, LDS transposition) --
// the T = 4 fc2 dX spent ~180 us per task output at stage 0.  Here it is a streaming elementwise kernel: a thread owns ONE 16-byte
// column chunk (its rp x 8 factor values live in registers) and walks the rows, 4 in flight; rounding as the tiled epilogue
// (fp32 sum -> T, then T * gelu'(h) -> T).
// ------------------------------------------------------------------------------------------------
struct RankOutParams {
    const void* Q;      // (M x ldq)
    const void* Acat;   // (R x K) row-major, unscaled (Q carries alpha)
    int64_t ldq, M;
    int K, n_t;
    int seg[MAXO], rp[MAXO];
    void* out[MAXO];
    const void* gate[MAXO];
};
template <typename T, bool GATE, int RP>
__global__ __launch_bounds__(256) void k_rank_out(const RankOutParams P) {
    constexpr int UNR = 4;
    const int t = blockIdx.y;
    const int nchunk = P.K >> 3;
    const int rpb = 256 / nchunk;  // rows per block sweep (nchunk <= 256)
    const int tid = threadIdx.x;
    if (tid >= rpb * nchunk) return;
    const int chunk = tid % nchunk, r0 = tid / nchunk;
    const int seg = P.seg[t];
    float a[RP][8];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(P.Acat) + (int64_t)(seg + j) * P.K + chunk * 8);
        VOps<T>::unpack(v, a[j]);
    }
    const T* q = reinterpret_cast<const T*>(P.Q) + seg;
    T* out = reinterpret_cast<T*>(P.out[t]);
    const T* gate = reinterpret_cast<const T*>(P.gate[t]);
    const int64_t step = (int64_t)gridDim.x * rpb;
    for (int64_t row = (int64_t)blockIdx.x * rpb + r0; row < P.M; row += UNR * step) {
        u32x4 qv[UNR][RP / 8], hv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t m = row + u * step;
            const int64_t mc = m < P.M ? m : P.M - 1;
#pragma unroll
            for (int w = 0; w < RP / 8; ++w) qv[u][w] = *reinterpret_cast<const u32x4*>(q + mc * P.ldq + w * 8);
            if constexpr (GATE) hv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gate + mc * P.K + chunk * 8));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t m = row + u * step;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int w = 0; w < RP / 8; ++w) {
                float qf[8];
                VOps<T>::unpack(qv[u][w], qf);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += qf[j] * a[w * 8 + j][e];
            }
            u32x4 o = VOps<T>::pack(acc);
            if constexpr (GATE) o = mtl_gelu_gate_pk4<T, false>(o, hv[u]);
            if (m < P.M) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(out + m * P.K + chunk * 8));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check_desc(const mtlora_linear_desc* d) {
    if (!d) return MTLORA_ERR_NULL;
    if (d->dtype != MTLORA_F32 && d->dtype != MTLORA_BF16 && d->dtype != MTLORA_F16) return MTLORA_ERR_DTYPE;
    if (d->M < 0 || d->K <= 0 || d->N <= 0 || d->T < 0 || d->T > MTLORA_MAX_TASKS || d->r_s < 0)
        return MTLORA_ERR_SHAPE;
    if (d->M >= ((int64_t)1 << 31)) return MTLORA_ERR_SHAPE;
    const int vec = d->dtype == MTLORA_F32 ? 4 : 8;
    if (d->K % vec || d->N % vec) return MTLORA_ERR_ALIGN;
    for (int t = 0; t < d->T; ++t)
        if (d->r_t[t] <= 0) return MTLORA_ERR_SHAPE;
    if (d->mode != 0 && d->mode != 1) return MTLORA_ERR_UNSUPPORTED;
    if (d->bwd_phase < 0 || d->bwd_phase > 2) return MTLORA_ERR_UNSUPPORTED;
    if (d->sel_stream < 0 || d->sel_stream > 1 || d->sel_dense < 0 || d->sel_dense > 4 || d->sel_tn < 0 || d->sel_tn > 2 || d->sel_projk < 0 ||
        d->sel_projk > 3 || d->max_cu < 0)
        return MTLORA_ERR_UNSUPPORTED;
    if (d->dropout_p < 0.f || d->dropout_p >= 1.f) return MTLORA_ERR_SHAPE;
    return MTLORA_OK;
}

static bool misaligned(const void* p) { return ((uintptr_t)p & 15u) != 0; }

// kernel selection of one call: a function of the descriptor alone (mtlora_linear_desc.sel_* / max_cu, ABI v6) -- the library reads
// no environment variables.  sp: wave-streaming family on; ntd / tn / projk: 0 never, 1 by heuristics, 2 whenever eligible (projk 3: k_pq
// whenever eligible).
struct Tune {
    int sp, ntd, tn, projk, max_cu;
};
static Tune make_tune(const mtlora_linear_desc* d) {
    auto tri = [](int v) { return v == 1 ? 0 : (v == 2 ? 2 : 1); };
    Tune t;
    t.sp = d->sel_stream == 1 ? 0 : 1;
    t.ntd = d->sel_dense >= 3 ? d->sel_dense : tri(d->sel_dense);  // (3: k_nte whenever eligible, 4: heuristics without k_nte)
    t.tn = tri(d->sel_tn);
    t.projk = d->sel_projk == 3 ? 3 : tri(d->sel_projk);  // (3: k_pq whenever eligible)
    t.max_cu = d->max_cu > 0 ? d->max_cu : 0;
    return t;
}
// per-device facts, cached (the only process-wide state next to the opt-in profiler): CU count and which kernels already had
// their dynamic-LDS limit raised on which device (hipFuncSetAttribute is per device)
constexpr int MTL_MAX_DEV = 64;
static int cur_dev() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < MTL_MAX_DEV ? dev : 0;
}
static int dev_num_cu() {
    static std::atomic<int> cache[MTL_MAX_DEV];
    const int dev = cur_dev();
    int cu = cache[dev].load(std::memory_order_relaxed);
    if (cu == 0) {
        cu = 256;
        (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (cu <= 0) cu = 256;
        cache[dev].store(cu, std::memory_order_relaxed);
    }
    return cu;
}
static int num_cu(const Tune& tu) {
    const int cu = dev_num_cu();
    return tu.max_cu > 0 && tu.max_cu < cu ? tu.max_cu : cu;
}
// raise the dynamic-LDS limit of `fn` to `bytes` once per device; false when the runtime refuses (the caller falls back / reports)
static bool raise_lds(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
    const int dev = cur_dev();
    if ((done.load(std::memory_order_relaxed) >> dev) & 1ull) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return true;
}
// (a refusal is not an error here: the launch that follows fails with an invalid-configuration error, which the entry point's
// MTL_CHECK_LAUNCH reports as MTLORA_ERR_HIP)
#define MTL_RAISE_LDS(KERNEL, BYTES)                                        \
    do {                                                                    \
        static std::atomic<unsigned long long> done__{0};                   \
        (void)raise_lds(done__, (const void*)(KERNEL), (int)(BYTES));       \
    } while (0)

// developer ablation bits (tools/nt_ablate.sh, tools/sp_ablate.sh): read from the environment ONLY in a -DMTL_NT_ABLATE=1 build
// (MTLORA_ABLATE=1 python -m mtlora_amd.csrc.build --force); the shipped library has neither the getenv nor the kernel branches
static int ablate_bits(const char* name) {
#if MTL_NT_ABLATE
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

// k_nte (two 4-wave workgroups per CU) instead of k_ntd / k_ntl?  Measured per shape (tools/ntd_ab.sh, profiles/r04_ntd_ab.txt): it wins
// where a tile spends a large share of its life outside the k loop -- short reductions (<= 12 k-steps of 64) with at least 1.5 tiles per
// CU -- and where k_ntd's one-workgroup-per-CU rounds are badly filled (< 70 %) while every CU still gets a tile; it loses on long
// reductions with few tiles (stage 3 dX: half the waves per CU).
static bool nte_prefer(int64_t tiles, int ksteps, int64_t cus) {
    const double eff_d = (double)tiles / (double)(mtl_ceil_div(tiles, cus) * cus);
    return (ksteps >= 4 && ksteps <= 12 && tiles * 2 >= 3 * cus) || (ksteps >= 6 && eff_d < 0.7 && tiles >= cus);
}

template <typename T>
static void launch_nt(const Tune& tu, const NtParams& P_in, hipStream_t s, int kind, double alg_bytes, double s8d_bytes = 0.0, double flops = 0.0) {
    NtParams P = P_in;
    P.dbg = ablate_bits("MTLORA_NT_DBG");
    mtl_prof_tag("M%lld K%d N%d ldL%lld no%d na%d nz%d", (long long)P.M, P.K, P.n_rows, (long long)P.ldL, P.n_out, P.n_act, P.nz);
    MtlProfScope prof(kind, alg_bytes, s, s8d_bytes, flops);
    int max_rows = P.n_rows;
    if (P.nz > 0) {
        max_rows = 0;
        for (int z = 0; z < P.nz; ++z) max_rows = P.zrows[z] > max_rows ? P.zrows[z] : max_rows;
    }
    const int64_t m_tiles = mtl_ceil_div(P.M, TILE);
    const int64_t n_tiles = mtl_ceil_div(max_rows, TILE);
    if (m_tiles * n_tiles == 0) return;
    dim3 g((unsigned)(m_tiles * n_tiles), 1, (unsigned)(P.nz > 0 ? P.nz : 1));
    int base_users = 0;
    for (int o = 0; o < P.n_out; ++o) base_users += P.out[o].use_base ? 1 : 0;
    const size_t lds = (size_t)STAGE_BYTES;
    // variant 0: several outputs share the base GEMM (MULTI: two accumulator sets, 4 waves); 1: multi-source activation (the dX
    // launch of a layer with task outputs); 2: one output, one source.  The 16-bit single-accumulator-set variants run 8 waves
    // per workgroup (<= 128 VGPRs, 4 waves per SIMD); MULTI and f32 (wider fragments) would spill at 128 and run 4.
    const int variant = base_users > 1 ? 0 : (P.n_act > 1 ? 1 : 2);
    constexpr bool W8 = sizeof(T) == 2;
    bool mlr = false, gated = false, acted = false;
    for (int o = 0; o < P.n_out; ++o) {
        mlr = mlr || P.out[o].mask_lr != 0;
        gated = gated || P.out[o].gate != nullptr;
        acted = acted || P.out[o].act != nullptr;
    }
    mlr = mlr && P.drop.enabled();
    if constexpr (std::is_same<T, bf16>::value) {
        // lean single-output bf16 launches (forward outputs, P / Q passes, rank-0 GEMMs, and the dX of layers without task
        // outputs -- masked rank part): the straight-line kernel
        // MFMA-dense launches (long reduction, enough tiles): k_ntd (dense.h).  MTLORA_NTD: 0 never, 2 whenever the shape allows
        const int ntd_mode = tu.ntd;  // 0 never, 1 by the heuristics below, 2 whenever the shape allows (the "[dense]" test variants)
        bool dense = false, use_e = false;
        int ksteps = 0;
        if (variant == 2 && P.n_out == 1 && P.nz == 0 && ntd_mode != 0 && P.act_mask == 0 && P.n_rows % 8 == 0 && P.n_rows >= 64 &&
            P.M < (int64_t)0x7FFFFF00 && P.out[0].ptr != nullptr && !(P.out[0].gate && P.out[0].act)) {
            const int seg = P.L ? P.out[0].seg_hi - P.out[0].seg_lo : 0;
            const int kk = (P.act[0] && P.out[0].use_base ? P.K : 0);
            ksteps = (seg + ND_KE - 1) / ND_KE + (kk + ND_KE - 1) / ND_KE;
            const int64_t tiles = mtl_ceil_div(P.M, ND_TM) * mtl_ceil_div(P.n_rows, ND_TN);
            const int64_t lim = ((int64_t)1 << 32) - 4096;
            const bool fits = P.M * P.ld_out * 2 < lim && P.M * P.ld_act * 2 < lim && P.M * P.ldL * 2 < lim && (int64_t)P.n_rows * P.ld_wgt * 2 < lim &&
                              (int64_t)P.n_rows * P.ldR * 2 < lim && P.K % 8 == 0 && seg % 8 == 0 && (P.ld_act % 8) == 0 && (P.ld_wgt % 8) == 0 &&
                              (P.ldL % 8) == 0 && (P.ldR % 8) == 0 && (P.L == nullptr || (P.out[0].seg_lo % 8) == 0);
            // where it wins (tools/ntd_ab.sh): a reduction of >= 6 k-tiles, residency rounds (one workgroup per CU) at least 70 % full or
            // a single round on at least half of the CUs, and no half-empty column tile
            const int64_t slots = num_cu(tu);
            const double eff = (double)tiles / (double)(mtl_ceil_div(tiles, slots) * slots);
            use_e = ntd_mode == 3 || (ntd_mode == 1 && nte_prefer(tiles, ksteps, slots) && (P.n_rows % ND_TN == 0 || P.n_rows > 2 * ND_TN));
            dense = fits && ksteps >= 1 &&
                    (ntd_mode == 2 || ntd_mode == 3 || use_e ||
                     (ksteps >= 6 && (eff >= 0.7 || (tiles <= slots && tiles >= slots / 2)) && (P.n_rows % ND_TN == 0 || P.n_rows > 2 * ND_TN)));
        }
        if (dense) {
            NlParams q;
            q.act = reinterpret_cast<const bf16*>(P.act[0]);
            q.wgt = reinterpret_cast<const bf16*>(P.wgt);
            q.L = reinterpret_cast<const bf16*>(P.L);
            q.Rm = reinterpret_cast<const bf16*>(P.Rm);
            q.out = reinterpret_cast<bf16*>(P.out[0].ptr);
            q.act2 = reinterpret_cast<bf16*>(P.out[0].act);
            q.gate = reinterpret_cast<const bf16*>(P.out[0].gate);
            q.bias = P.bias;
            q.alpha = P.alpha;
            q.ld_act = P.ld_act;
            q.ld_wgt = P.ld_wgt;
            q.ldL = P.ldL;
            q.ldR = P.ldR;
            q.ld_out = P.ld_out;
            q.M = (int)P.M;
            q.n_rows = P.n_rows;
            q.K = P.act[0] ? P.K : 0;
            q.seg_lo = P.L ? P.out[0].seg_lo : 0;
            q.seg_hi = P.L ? P.out[0].seg_hi : 0;
            const int64_t mt = mtl_ceil_div(P.M, ND_TM), nt = mtl_ceil_div(P.n_rows, ND_TN);
            q.n_tiles = (int)nt;
            q.nt_magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)nt) + 1u;
            const uint32_t nwg = (uint32_t)(mt * nt);
            q.q8 = nwg / 8u;
            q.r8 = nwg % 8u;
            q.act_mask = 0;
            q.use_base = P.out[0].use_base;
            q.dbg = P.dbg;
            q.pad_ = 0;
            q.drop = P.drop;
            const bool ml0 = P.out[0].mask_lr != 0 && P.drop.enabled() && q.seg_hi > q.seg_lo;
            const uint32_t grid = nwg < (uint32_t)num_cu(tu) ? nwg : (uint32_t)num_cu(tu);
#define MTL_NTD_GO(AC, ML, GA)                                                                \
    do {                                                                                      \
        MTL_RAISE_LDS((k_ntd<AC, ML, GA>), SP_LDS_MAX);                                       \
        hipLaunchKernelGGL((k_ntd<AC, ML, GA>), dim3(grid), dim3(512), (size_t)ND_LDS, s, q); \
    } while (0)
            // k_nte: the same tile with two 4-wave workgroups per CU (dense.h); ntd_mode 3 forces it, 2 forces k_ntd
            if (use_e) {
                const uint32_t ge = nwg < 2u * (uint32_t)num_cu(tu) ? nwg : 2u * (uint32_t)num_cu(tu);
#define MTL_NTE_GO(AC, ML, GA)                                                                \
    do {                                                                                      \
        MTL_RAISE_LDS((k_nte<AC, ML, GA>), SP_LDS_MAX);                                       \
        hipLaunchKernelGGL((k_nte<AC, ML, GA>), dim3(ge), dim3(256), (size_t)NE_LDS, s, q);   \
    } while (0)
                if (q.act2)
                    MTL_NTE_GO(true, false, false);
                else if (q.gate && ml0)
                    MTL_NTE_GO(false, true, true);
                else if (q.gate)
                    MTL_NTE_GO(false, false, true);
                else if (ml0)
                    MTL_NTE_GO(false, true, false);
                else
                    MTL_NTE_GO(false, false, false);
#undef MTL_NTE_GO
                return;
            }
            if (q.act2)
                MTL_NTD_GO(true, false, false);
            else if (q.gate && ml0)
                MTL_NTD_GO(false, true, true);
            else if (q.gate)
                MTL_NTD_GO(false, false, true);
            else if (ml0)
                MTL_NTD_GO(false, true, false);
            else
                MTL_NTD_GO(false, false, false);
#undef MTL_NTD_GO
            return;
        }
        if (variant == 2 && P.n_out == 1 && P.out[0].gate == nullptr && P.nz == 0 && P.M < (int64_t)0x7FFFFF00 &&
            m_tiles * n_tiles < ((int64_t)1 << 28) && P.n_rows >= 8 && P.n_rows % 8 == 0) {
            NlParams q;
            q.act = reinterpret_cast<const bf16*>(P.act[0]);
            q.wgt = reinterpret_cast<const bf16*>(P.wgt);
            q.L = reinterpret_cast<const bf16*>(P.L);
            q.Rm = reinterpret_cast<const bf16*>(P.Rm);
            q.out = reinterpret_cast<bf16*>(P.out[0].ptr);
            q.act2 = reinterpret_cast<bf16*>(P.out[0].act);
            q.gate = nullptr;
            q.bias = P.bias;
            q.alpha = P.alpha;
            q.ld_act = P.ld_act;
            q.ld_wgt = P.ld_wgt;
            q.ldL = P.ldL;
            q.ldR = P.ldR;
            q.ld_out = P.ld_out;
            q.M = (int)P.M;
            q.n_rows = P.n_rows;
            q.K = P.act[0] ? P.K : 0;
            q.seg_lo = P.L ? P.out[0].seg_lo : 0;
            q.seg_hi = P.L ? P.out[0].seg_hi : 0;
            // tile width: 128 columns, or 192 when that saves residency rounds (512 workgroup slots at 128 columns, 2 per CU either way;
            // a 192-wide tile is 1.5x the work)
            const int64_t t128 = m_tiles * mtl_ceil_div(P.n_rows, 128), t192 = m_tiles * mtl_ceil_div(P.n_rows, 192);
            const int64_t slots = 2 * (int64_t)dev_num_cu();
            const double c128 = (double)mtl_ceil_div(t128, slots), c192 = 1.5 * (double)mtl_ceil_div(t192, slots);
            const bool wide = !q.act2 && P.n_rows >= 192 && c192 < c128 - 0.01;
            const int64_t nt = wide ? mtl_ceil_div(P.n_rows, 192) : n_tiles;
            q.n_tiles = (int)nt;
            q.nt_magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)nt) + 1u;
            const uint32_t nwg = (uint32_t)(m_tiles * nt);
            q.q8 = nwg / 8u;
            q.r8 = nwg % 8u;
            q.act_mask = P.act_mask;
            q.use_base = P.out[0].use_base;
            q.dbg = P.dbg;
            q.pad_ = 0;
            q.drop = P.drop;
            if (q.out == nullptr) return;
            const bool ml0 = P.out[0].mask_lr != 0 && P.drop.enabled() && q.seg_hi > q.seg_lo;
            constexpr size_t LDS192 = (size_t)(192 + 128) * LDSB;
#define MTL_NTL_GO(AC, ML, SNV, LDSV)                                                         \
    do {                                                                                      \
        if ((LDSV) > 64 * 1024) MTL_RAISE_LDS((k_ntl<AC, ML, SNV>), 160 * 1024 - 512);        \
        hipLaunchKernelGGL((k_ntl<AC, ML, SNV>), dim3(nwg), dim3(512), (size_t)(LDSV), s, q); \
    } while (0)
            if (q.act2)
                MTL_NTL_GO(true, false, 2, STAGE_BYTES);
            else if (wide && ml0)
                MTL_NTL_GO(false, true, 3, LDS192);
            else if (wide)
                MTL_NTL_GO(false, false, 3, LDS192);
            else if (ml0)
                MTL_NTL_GO(false, true, 2, STAGE_BYTES);
            else
                MTL_NTL_GO(false, false, 2, STAGE_BYTES);
#undef MTL_NTL_GO
            return;
        }
    }
#define MTL_NT_GO(MU, MSRC, ML, GA, AC)                                                                   \
    do {                                                                                                \
        if (W8 && !(MU))                                                                                \
            hipLaunchKernelGGL((k_nt<T, false, MSRC, ML, 8, GA, AC>), g, dim3(512), lds, s, P);         \
        else                                                                                            \
            hipLaunchKernelGGL((k_nt<T, MU, MSRC, ML, 4, GA, AC>), g, dim3(256), lds, s, P);            \
    } while (0)
    if (acted) {  // forward outputs with the GELU second output (fc1 of the Mlp): lean or MULTI
        if (variant == 0)
            MTL_NT_GO(true, false, false, false, true);
        else
            MTL_NT_GO(false, false, false, false, true);
    } else if (gated) {  // dX * gelu'(h) epilogue: the lean (single accumulator set) variants
        if (variant == 1)
            MTL_NT_GO(false, true, true, true, false);
        else if (mlr)
            MTL_NT_GO(false, false, true, true, false);
        else
            MTL_NT_GO(false, false, false, true, false);
    } else if (variant == 0) {
        MTL_NT_GO(true, false, false, false, false);
    } else if (variant == 1) {
        MTL_NT_GO(false, true, true, false, false);  // multi-source = the dX launch
    } else if (mlr) {
        MTL_NT_GO(false, false, true, false, false);
    } else {
        MTL_NT_GO(false, false, false, false, false);
    }
#undef MTL_NT_GO
}

// ---- wave-streaming projection (k_sp_proj, stream.h): the P = alpha D(X) A^T / Q = alpha dY B passes
// fills q.n_blk_total / n_slabs / n_items and returns the ring depth (0: not eligible).  Sources must be set.
template <typename T>
static int sp_proj_plan(const Tune& tu, SpProjParams& q, int& ch) {
    if (sizeof(T) != 2 || tu.sp == 0 || q.M <= 0 || q.n_src <= 0) return 0;
    ch = q.K % 96 == 0 ? 96 : (q.K % 64 == 0 ? 64 : 0);
    if (ch == 0 || q.M >= ((int64_t)1 << 31) - 64) return 0;
    q.n_blk_total = (q.Rw + 31) / 32;
    for (int s = 0; s < q.n_src; ++s) {
        if (q.src[s].n_blk > SP_MAXB || q.src[s].n_blk <= 0) return 0;
        if (((uintptr_t)q.src[s].act & 15u) != 0) return 0;
    }
    if ((q.ld_out % 8) != 0 || ((uintptr_t)q.out & 15u) != 0 || ((uintptr_t)q.wproj & 15u) != 0) return 0;
    if (q.M * q.ld_out * 2 >= ((int64_t)1 << 32) - 64) return 0;  // 32-bit store offsets (buffer descriptor)
    q.n_slabs = (int)mtl_ceil_div(q.M, 32);
    if ((int64_t)q.n_slabs * q.n_src >= ((int64_t)1 << 30)) return 0;
    q.n_items = q.n_slabs * q.n_src;
    const int64_t wbytes = (int64_t)q.n_blk_total * 32 * q.K * 2;
    const int64_t slot = 32 * ch * 2;
    for (int ns = 3; ns >= 1; --ns)
        if (wbytes + (int64_t)SP_WAVES * ns * slot <= SP_LDS_MAX) return ns;
    return 0;
}
template <typename T>
static void launch_sp_proj(const Tune& tu, const SpProjParams& q, int ch, int ns, hipStream_t s, int kind, double alg_bytes, double s8d, double flops) {
    mtl_prof_tag("sp_proj M%lld K%d R%d src%d ch%d ns%d", (long long)q.M, q.K, q.Rw, q.n_src, ch, ns);
    MtlProfScope prof(kind, alg_bytes, s, s8d, flops);
    const size_t lds = (size_t)q.n_blk_total * 32 * q.K * 2 + (size_t)SP_WAVES * ns * 32 * ch * 2;
    const int per_cu = lds * 2 <= (size_t)SP_LDS_MAX ? 2 : 1;
    int64_t wgs = mtl_ceil_div(q.n_items, SP_WAVES);
    if (wgs > (int64_t)num_cu(tu) * per_cu) wgs = (int64_t)num_cu(tu) * per_cu;
#define MTL_SP_PROJ(CHV, NSV)                                                                              \
    do {                                                                                                   \
        MTL_RAISE_LDS((k_sp_proj<T, CHV, NSV>), SP_LDS_MAX);                                               \
        hipLaunchKernelGGL((k_sp_proj<T, CHV, NSV>), dim3((unsigned)wgs), dim3(64 * SP_WAVES), lds, s, q); \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        if (ch == 96) {
            if (ns == 3) MTL_SP_PROJ(96, 3);
            else if (ns == 2) MTL_SP_PROJ(96, 2);
            else MTL_SP_PROJ(96, 1);
        } else {
            if (ns == 3) MTL_SP_PROJ(64, 3);
            else if (ns == 2) MTL_SP_PROJ(64, 2);
            else MTL_SP_PROJ(64, 1);
        }
    }
#undef MTL_SP_PROJ
}

// ---- k_pq (pq.h): the P / Q passes whose projection rows do not fit in LDS: 64- or 128-row tiles with a deep LDS-DMA ring, one grid
// slice (blockIdx.z) per source with its column segment.  Fills `pq`, returns the row blocks per wave (1: 64-row tiles, 2: 128-row
// tiles; 0: not eligible).
template <typename T>
static int pq_plan(const Tune& tu, const SpProjParams& q, PqParams& pq) {
    if (sizeof(T) != 2 || tu.sp == 0 || tu.projk == 0 || tu.projk == 2 || q.M <= 0 || q.n_src < 1 || q.n_src > MAXO) return 0;
    if (q.Rw % 8 != 0 || q.K % 8 != 0 || q.K < 32 || (q.ld_out % 8) != 0) return 0;
    if ((((uintptr_t)q.wproj | (uintptr_t)q.out) & 15u) != 0) return 0;
    if (q.M >= ((int64_t)1 << 31) - 256 || q.M * q.ld_out * 2 >= ((int64_t)1 << 32) - 64) return 0;
    int wmax = 0;
    for (int s = 0; s < q.n_src; ++s) {
        const SpSrc& ss = q.src[s];
        if (((uintptr_t)ss.act & 15u) != 0 || ss.col_lo % 8 != 0 || ss.col_hi % 8 != 0 || ss.col_hi <= ss.col_lo || ss.col_hi > q.Rw) return 0;
        if (ss.col_hi - ss.col_lo > 1024) return 0;  // (<= 8 column tiles per source)
        pq.src[s].act = ss.act;
        pq.src[s].col_lo = ss.col_lo;
        pq.src[s].col_hi = ss.col_hi;
        pq.src[s].mask = ss.mask;
        pq.src[s].pad_ = 0;
        wmax = ss.col_hi - ss.col_lo > wmax ? ss.col_hi - ss.col_lo : wmax;
    }
    pq.n_src = q.n_src;
    pq.wproj = q.wproj;
    pq.out = q.out;
    pq.ld_out = q.ld_out;
    pq.M = (int)q.M;
    pq.K = q.K;
    pq.R = q.Rw;
    pq.drop = q.drop;
    // 64-row tiles while the launch is about one residency round (a workgroup per CU), 128-row tiles beyond
    const int64_t per_row_tile = (int64_t)q.n_src * mtl_ceil_div(wmax, wmax > 64 ? 128 : 64);
    return mtl_ceil_div(q.M, 64) * per_row_tile * 4 <= (int64_t)num_cu(tu) * 5 ? 1 : 2;
}
static bool pq_one_round(const Tune& tu, const PqParams& pq, int mb) {
    return pq.n_src == 1 && mtl_ceil_div(pq.M, 64 * mb) * 4 <= (int64_t)num_cu(tu) * 5;
}
template <typename T>
static void launch_pq(const PqParams& pq, int mb, hipStream_t s, int kind, double alg_bytes, double s8d, double flops) {
    int wmax = 0, any_mask = 0;
    for (int i = 0; i < pq.n_src; ++i) {
        wmax = pq.src[i].col_hi - pq.src[i].col_lo > wmax ? pq.src[i].col_hi - pq.src[i].col_lo : wmax;
        any_mask |= pq.src[i].mask;
    }
    mtl_prof_tag("pq M%d K%d R%d src%d w%d mb%d mask%d", pq.M, pq.K, pq.R, pq.n_src, wmax, mb, any_mask);
    MtlProfScope prof(kind, alg_bytes, s, s8d, flops);
    const bool wide = wmax > 64;
    const dim3 grid((unsigned)mtl_ceil_div(pq.M, 64 * mb), (unsigned)mtl_ceil_div(wmax, wide ? 128 : 64), (unsigned)pq.n_src);
#define MTL_PQ(WMV, NBV, NSTV, KSV)                                                                               \
    do {                                                                                                          \
        constexpr size_t lds = (size_t)NSTV * (32 * WMV + 32 * NBV) * 64;                                         \
        MTL_RAISE_LDS((k_pq<T, WMV, NBV, NSTV, KSV>), SP_LDS_MAX);                                                \
        hipLaunchKernelGGL((k_pq<T, WMV, NBV, NSTV, KSV>), grid, dim3(256), lds, s, pq);                          \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        const bool ksp = any_mask != 0 && pq.drop.enabled();  // (masked: every activation fragment hashed by one wave)
        if (mb == 1 && wide && ksp) MTL_PQ(2, 4, 6, true);   // 64 x 128: 12 KB stages, 60 KB in flight
        else if (mb == 1 && wide) MTL_PQ(2, 4, 6, false);
        else if (mb == 1 && ksp) MTL_PQ(2, 2, 8, true);      // 64 x 64:   8 KB stages, 56 KB in flight
        else if (mb == 1) MTL_PQ(2, 2, 8, false);
        else if (wide) MTL_PQ(4, 4, 5, false);               // 128 x 128: 16 KB stages, 64 KB in flight
        else MTL_PQ(4, 2, 6, false);                         // 128 x 64: 12 KB stages, 60 KB in flight
    }
#undef MTL_PQ
}

// ---- k_sp_projk (stream.h): the P / Q passes whose projection rows do not fit in LDS, for small row counts (one work item per
// workgroup, reduction split over its waves).  Same parameter block as k_sp_proj; returns false when the shape is not eligible.
template <typename T>
static bool sp_projk_plan(const Tune& tu, SpProjParams& q, int& ch) {
    if (sizeof(T) != 2 || tu.sp == 0 || q.M <= 0 || q.n_src <= 0) return false;
    const int mode = tu.projk;
    if (mode == 0) return false;
    ch = q.K % 96 == 0 ? 96 : (q.K % 64 == 0 ? 64 : 0);
    if (ch == 0 || q.M >= ((int64_t)1 << 31) - 64) return false;
    q.n_blk_total = (q.Rw + 31) / 32;
    for (int s = 0; s < q.n_src; ++s) {
        if (q.src[s].n_blk > SP_MAXB || q.src[s].n_blk <= 0) return false;
        if (((uintptr_t)q.src[s].act & 15u) != 0) return false;
    }
    if ((q.ld_out % 8) != 0 || ((uintptr_t)q.out & 15u) != 0 || ((uintptr_t)q.wproj & 15u) != 0) return false;
    if (q.M * q.ld_out * 2 >= ((int64_t)1 << 32) - 64) return false;
    q.n_slabs = (int)mtl_ceil_div(q.M, 32);
    q.n_items = q.n_slabs * q.n_src;
    // every item re-reads the projection rows from L2 and pays three barriers: it wins while the whole launch is ONE residency round
    // (stage 3: 196 slabs; 42 - 62 us -> 20 - 24 us), ties at two to three rounds and loses beyond (tools/projk_ab.sh)
    return mode == 2 || (mode == 1 && (int64_t)q.n_items <= (int64_t)num_cu(tu) && q.K / ch >= SP_WAVES);
}
template <typename T>
static void launch_sp_projk(const Tune& tu, const SpProjParams& q, int ch, hipStream_t s, int kind, double alg_bytes, double s8d, double flops) {
    mtl_prof_tag("sp_projk M%lld K%d R%d src%d ch%d", (long long)q.M, q.K, q.Rw, q.n_src, ch);
    MtlProfScope prof(kind, alg_bytes, s, s8d, flops);
    int64_t wgs = q.n_items;
    if (wgs > (int64_t)num_cu(tu)) wgs = num_cu(tu);
#define MTL_SP_PROJK(CHV, NSLV)                                                                                   \
    do {                                                                                                          \
        constexpr size_t slots = (size_t)SP_WAVES * NSLV * 32 * CHV * 2, red = (size_t)SP_WAVES * SP_MAXB * 4096; \
        constexpr size_t lds = slots > red ? slots : red;                                                         \
        MTL_RAISE_LDS((k_sp_projk<T, CHV, NSLV>), SP_LDS_MAX);                                                    \
        hipLaunchKernelGGL((k_sp_projk<T, CHV, NSLV>), dim3((unsigned)wgs), dim3(64 * SP_WAVES), lds, s, q);      \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        if (ch == 96)
            MTL_SP_PROJK(96, 3);
        else
            MTL_SP_PROJK(64, 4);
    }
#undef MTL_SP_PROJK
}

template <typename T>
static void launch_sp_projsum(const Tune& tu, const SpProjParams& q, int ch, int ns, T* gsum, hipStream_t s, int kind, double alg_bytes,
                              double s8d, double flops) {
    mtl_prof_tag("sp_projsum M%lld K%d R%d src%d ch%d ns%d", (long long)q.M, q.K, q.Rw, q.n_src, ch, ns);
    MtlProfScope prof(kind, alg_bytes, s, s8d, flops);
    const size_t lds = (size_t)q.n_blk_total * 32 * q.K * 2 + (size_t)SP_WAVES * ns * 32 * ch * 2;
    int64_t wgs = mtl_ceil_div(q.n_slabs, SP_WAVES);
    if (wgs > (int64_t)num_cu(tu)) wgs = num_cu(tu);
#define MTL_SP_PS(CHV, NSV)                                                                                         \
    do {                                                                                                            \
        MTL_RAISE_LDS((k_sp_projsum<T, CHV, NSV>), SP_LDS_MAX);                                                     \
        hipLaunchKernelGGL((k_sp_projsum<T, CHV, NSV>), dim3((unsigned)wgs), dim3(64 * SP_WAVES), lds, s, q, gsum); \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        if (ch == 96) {
            if (ns >= 2) MTL_SP_PS(96, 2);
            else MTL_SP_PS(96, 1);
        } else {
            if (ns >= 2) MTL_SP_PS(64, 2);
            else MTL_SP_PS(64, 1);
        }
    }
#undef MTL_SP_PS
}

// ---- fused wave-streaming MTLoRALinear launch, activation-resident form (k_sp_xres, stream.h).  Fills the plan fields of q
// (n_parts, blk_per_part, n_slabs, estep) and the launch geometry; false: not eligible (weights do not fit / unsupported shape).
struct SpXresPlan {
    int ch, nkc, nrb;
    bool stg;
    unsigned grid;
    size_t lds;
};
// persistent grid of a fused wave-streaming launch: `parts` column parts x slab groups.  Normally the parts of one slab group sit on
// one XCD (blockIdx b -> XCD b % 8), so the groups come in eights (xsh = 3).  With fewer than 8 * parts CUs to use (desc.max_cu: the
// "[persist]" tests) the eight-fold grouping is dropped (xsh = 0): part = b % parts, group = b / parts.
static unsigned sp_lin_grid(const Tune& tu, int parts, int n_slabs, int& xsh) {
    const int64_t cu = num_cu(tu);
    const int64_t by_slabs = mtl_ceil_div(n_slabs, SP_WAVES);
    if (cu >= 8 * (int64_t)parts) {
        xsh = 3;
        int64_t g8 = mtl_ceil_div(by_slabs, 8);
        const int64_t g8_max = cu / (8 * parts);
        if (g8 > g8_max) g8 = g8_max;
        return (unsigned)(8 * parts * g8);
    }
    xsh = 0;
    int64_t g = cu / parts > 0 ? cu / parts : 1;
    if (g > by_slabs) g = by_slabs;
    return (unsigned)(parts * g);
}
static int sp_stg_mode() {  // developer switch (ablation builds only): MTLORA_SP_STG=0 direct row-per-lane stores, 1 staged stores whenever they fit
#if MTL_NT_ABLATE
    static const int m = [] { const char* e = getenv("MTLORA_SP_STG"); return e ? atoi(e) : -1; }();
    return m;
#else
    return -1;
#endif
}
template <typename T>
static bool sp_xres_plan(const Tune& tu, SpLinParams& q, int K, SpXresPlan& pl) {
    if (sizeof(T) != 2 || tu.sp == 0 || q.M <= 0 || q.M >= ((int64_t)1 << 31) - 64) return false;
    if (K == 96 || K == 192) {
        pl.ch = 96;
        pl.nkc = K / 96;
    } else if (K == 64 || K == 128) {
        pl.ch = 64;
        pl.nkc = K / 64;
    } else {
        return false;
    }
    if (q.R <= 0 || q.R > 128 || (q.R % 16) != 0) return false;
    pl.nrb = q.R <= 32 ? 1 : (q.R <= 64 ? 2 : 4);
    q.estep = 2 * ((q.R + 31) / 32);
    if ((q.n_cols % 8) != 0 || (q.ld_out % 8) != 0 || (q.ldp % 8) != 0) return false;
    if (q.M * q.ld_out * 2 >= ((int64_t)1 << 32) - 64 || q.M * q.ldp * 2 >= ((int64_t)1 << 32) - 64) return false;  // 32-bit store offsets
    const void* ptrs[] = {q.act, q.w, q.proj, q.expand, q.out, q.out2, q.pout, q.gate};
    for (const void* pp : ptrs)
        if (((uintptr_t)pp & 15u) != 0) return false;
    const int nb_all = (q.n_cols + 31) / 32;
    const int64_t slot = (int64_t)SP_WAVES * 32 * pl.ch * 2;
    auto fit = [&](bool stg, int& bpp_out, size_t& lds_out) -> int {
        const int64_t fixed = (int64_t)pl.nrb * 32 * K * 2 + slot + (stg ? (int64_t)SP_WAVES * SP_STG : 0);
        for (int np = 1; np <= nb_all && np <= 16; ++np) {
            const int bpp = (nb_all + np - 1) / np;
            if ((bpp * (np - 1)) >= nb_all) continue;  // an empty last part
            const int64_t need = fixed + (int64_t)bpp * (32 * K * 2 + 64 * pl.nrb * 32 * 2 / 2 + 128);
            if (need <= SP_LDS_MAX) {
                bpp_out = bpp;
                lds_out = (size_t)need;
                return np;
            }
        }
        return 0;
    };
    int bpp_d = 0, bpp_s = 0;
    size_t lds_d = 0, lds_s = 0;
    const int np_d = fit(false, bpp_d, lds_d), np_s = fit(true, bpp_s, lds_s);
    const int mode = sp_stg_mode();
    bool stg = np_s > 0 && (np_d == 0 || np_s <= np_d + 1);
    if (mode == 0 && np_d > 0) stg = false;
    if (mode == 1 && np_s > 0) stg = true;
    const int parts = stg ? np_s : np_d;
    if (parts == 0) return false;
    pl.stg = stg;
    q.blk_per_part = stg ? bpp_s : bpp_d;
    pl.lds = stg ? lds_s : lds_d;
    q.n_parts = parts;
    q.dbg = ablate_bits("MTLORA_SP_DBG");
    q.n_slabs = (int)mtl_ceil_div(q.M, 32);
    pl.grid = sp_lin_grid(tu, parts, q.n_slabs, q.xsh);
    return true;
}
template <typename T>
static void launch_sp_xres(const SpLinParams& q, const SpXresPlan& pl, bool act, hipStream_t s, int kind, double alg_bytes, double s8d,
                           double flops, bool gate = false) {
    mtl_prof_tag("sp_xres M%lld K%d N%d R%d parts%d stg%d", (long long)q.M, pl.ch * pl.nkc, q.n_cols, q.R, q.n_parts, pl.stg ? 1 : 0);
    MtlProfScope prof(kind, alg_bytes, s, s8d, flops);
#define MTL_SP_X1(CHV, NKCV, NRBV, ACTV, STGV, GAV)                                                                             \
    do {                                                                                                                        \
        MTL_RAISE_LDS((k_sp_xres<T, CHV, NKCV, NRBV, ACTV, STGV, GAV>), SP_LDS_MAX);                                            \
        hipLaunchKernelGGL((k_sp_xres<T, CHV, NKCV, NRBV, ACTV, STGV, GAV>), dim3(pl.grid), dim3(64 * SP_WAVES), pl.lds, s, q); \
    } while (0)
#define MTL_SP_X(CHV, NKCV, NRBV, ACTV, STGV)                                    \
    do {                                                                         \
        if (!(ACTV) && gate) MTL_SP_X1(CHV, NKCV, NRBV, false, STGV, true);      \
        else MTL_SP_X1(CHV, NKCV, NRBV, ACTV, STGV, false);                      \
    } while (0)
#define MTL_SP_X_S(CHV, NKCV, NRBV, ACTV)                        \
    do {                                                         \
        if (pl.stg) MTL_SP_X(CHV, NKCV, NRBV, ACTV, true);       \
        else MTL_SP_X(CHV, NKCV, NRBV, ACTV, false);             \
    } while (0)
#define MTL_SP_X_R(CHV, NKCV, ACTV)                            \
    do {                                                       \
        if (pl.nrb == 1) MTL_SP_X_S(CHV, NKCV, 1, ACTV);       \
        else if (pl.nrb == 2) MTL_SP_X_S(CHV, NKCV, 2, ACTV);  \
        else MTL_SP_X_S(CHV, NKCV, 4, ACTV);                   \
    } while (0)
#define MTL_SP_X_K(ACTV)                                                  \
    do {                                                                  \
        if (pl.ch == 96 && pl.nkc == 1) MTL_SP_X_R(96, 1, ACTV);          \
        else if (pl.ch == 96) MTL_SP_X_R(96, 2, ACTV);                    \
        else if (pl.nkc == 1) MTL_SP_X_R(64, 1, ACTV);                    \
        else MTL_SP_X_R(64, 2, ACTV);                                     \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        if (act) MTL_SP_X_K(true);
        else MTL_SP_X_K(false);
    }
#undef MTL_SP_X_K
#undef MTL_SP_X_R
#undef MTL_SP_X_S
#undef MTL_SP_X
#undef MTL_SP_X1
}

// ---- fused wave-streaming launch, accumulator-resident form (k_sp_ares, stream.h): few output columns, any reduction length
struct SpAresPlan {
    int ch, nob, nrb;
    bool expg;  // expansion fragments from global memory instead of LDS (k_sp_ares<..., EXPG>)
    unsigned grid;
    size_t lds;
};
template <typename T>
static bool sp_ares_plan(const Tune& tu, SpLinParams& q, int Kred, SpAresPlan& pl) {
    if (sizeof(T) != 2 || tu.sp == 0 || q.M <= 0 || q.M >= ((int64_t)1 << 31) - 64) return false;
    pl.ch = Kred % 96 == 0 ? 96 : (Kred % 64 == 0 ? 64 : 0);
    if (pl.ch == 0) return false;
    pl.nob = pl.ch == 96 ? 3 : 4;
    pl.expg = false;
    if (q.R <= 0 || q.R > 128 || (q.R % 16) != 0) return false;
    pl.nrb = q.R <= 64 ? 2 : 4;
    q.estep = 2 * ((q.R + 31) / 32);
    q.estep2 = Kred;
    if ((q.n_cols % 8) != 0 || (q.ld_out % 8) != 0 || (q.ldp % 8) != 0) return false;
    if (q.M * q.ld_out * 2 >= ((int64_t)1 << 32) - 64 || q.M * q.ldp * 2 >= ((int64_t)1 << 32) - 64) return false;
    const void* ptrs[] = {q.act, q.w, q.proj, q.expand, q.out, q.pout};
    for (const void* pp : ptrs)
        if (((uintptr_t)pp & 15u) != 0) return false;
    const int nb_all = (q.n_cols + 31) / 32;
    const int parts = (nb_all + pl.nob - 1) / pl.nob;
    if (parts > 2) return false;  // every part re-reads the activation: only worth it for narrow outputs
    const int64_t kst = Kred / 16;
    int64_t need = (int64_t)pl.nob * kst * 1024 + (int64_t)pl.nrb * kst * 1024 + (int64_t)pl.nob * 2 * pl.nrb * 1024 + pl.nob * 128 +
                   (int64_t)SP_WAVES * 32 * pl.ch * 2;
    if (need > SP_LDS_MAX && nb_all <= 3 && Kred % 64 == 0 && pl.nrb == 2) {
        // a long reduction with a 64-wide rank (stage-0 fc2 forward / fc1 dX: 384 -> 96): 64-wide slots, three output blocks and the
        // expansion fragments left in global memory (k_sp_ares<T, 64, 3, 2, EXPG>)
        pl.ch = 64;
        pl.nob = 3;
        pl.expg = true;
        need = (int64_t)pl.nob * kst * 1024 + (int64_t)pl.nrb * kst * 1024 + pl.nob * 128 + (int64_t)SP_WAVES * 32 * pl.ch * 2;
    }
    if (need > SP_LDS_MAX) return false;
    pl.lds = (size_t)need;
    q.n_parts = parts;
    q.blk_per_part = pl.nob;
    q.dbg = ablate_bits("MTLORA_SP_DBG");
    q.n_slabs = (int)mtl_ceil_div(q.M, 32);
    pl.grid = sp_lin_grid(tu, parts, q.n_slabs, q.xsh);
    return true;
}
template <typename T>
static void launch_sp_ares(const SpLinParams& q, const SpAresPlan& pl, hipStream_t s, int kind, double alg_bytes, double s8d, double flops) {
    mtl_prof_tag("sp_ares M%lld Kred%d N%d R%d parts%d", (long long)q.M, q.estep2, q.n_cols, q.R, q.n_parts);
    MtlProfScope prof(kind, alg_bytes, s, s8d, flops);
#define MTL_SP_A(CHV, NOBV, NRBV)                                                                              \
    do {                                                                                                       \
        MTL_RAISE_LDS((k_sp_ares<T, CHV, NOBV, NRBV>), SP_LDS_MAX);                                            \
        hipLaunchKernelGGL((k_sp_ares<T, CHV, NOBV, NRBV>), dim3(pl.grid), dim3(64 * SP_WAVES), pl.lds, s, q); \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        if (pl.expg) {
            MTL_RAISE_LDS((k_sp_ares<T, 64, 3, 2, true>), SP_LDS_MAX);
            hipLaunchKernelGGL((k_sp_ares<T, 64, 3, 2, true>), dim3(pl.grid), dim3(64 * SP_WAVES), pl.lds, s, q);
        } else if (pl.ch == 96) {
            if (pl.nrb == 2) MTL_SP_A(96, 3, 2);
            else MTL_SP_A(96, 3, 4);
        } else {
            if (pl.nrb == 2) MTL_SP_A(64, 4, 2);
            else MTL_SP_A(64, 4, 4);
        }
    }
#undef MTL_SP_A
}

// k_pack of one layer into the packed-factor region at `pk` (the head of a forward's ctx buffer, or the caller's persistent
// buffer of mtlora_linear_pack)
// ---- wave-streaming factor gradients (k_sp_tn, stream.h).  Geometry is a function of the layer shape only (the scratch size
// must not depend on the device): part width 32 nb columns (nb = 4 when K and N are multiples of 128, 3 when multiples of 96),
// ~SP_TN_WGS workgroups dealt to the pps in row groups of 8.
constexpr int SP_TN_WGS = 256;
static int sp_tn_nb(const mtlora_linear_desc* d) {
    if (mtl_elem_size(d->dtype) != 2 || d->M <= 0 || d->M >= ((int64_t)1 << 31) - 64) return 0;
    if (d->K % 128 == 0 && d->N % 128 == 0) return 4;
    if (d->K % 96 == 0 && d->N % 96 == 0) return 3;
    return 0;
}
static int sp_tn_groups(const Tune& tu, int n_pp, int64_t M) {
    int G = SP_TN_WGS / (n_pp > 0 ? n_pp : 1);  // one resident round of workgroups, every workgroup the same number of rows
    if (tu.max_cu > 0) {  // (tests: as if the device had max_cu CUs -- fewer row groups, more slabs per wave)
        const int cap = tu.max_cu / (n_pp > 0 ? n_pp : 1);
        G = G < cap ? G : (cap > 0 ? cap : 1);
    }
    const int64_t by_rows = mtl_ceil_div(mtl_ceil_div(M, 32), SP_WAVES);  // no more waves than slabs
    if (G > by_rows) G = (int)by_rows;
    return G < 1 ? 1 : G;
}
// blockIdx -> (pp, row group).  Placement units are dealt largest first, each to the XCD with the least load so far.  coarse: a unit
// is every pp that reads the same narrow MATRIX for one row group (all dA problems read Q, all dB problems P: their column windows
// share cache lines); fine: one (problem, narrow tile).  Coarse is kept when no XCD gets more than its 32 CUs' worth of workgroups.
// Returns the grid size (0: does not fit the table).
static unsigned sp_tn_map_units(SpTnParams& q, bool coarse, int& max_load) {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int CAP = SP_TN_MAP / 8;
    for (int i = 0; i < SP_TN_MAP; ++i) q.map[i] = 0xFFFFFFFFu;
    // unit list: [first problem, last problem] ranges over a problem order in which equal narrow matrices are adjacent
    int order[2 * MAXO], n = q.n_prob;
    for (int i = 0; i < n; ++i) order[i] = i;
    auto size_of = [&](int i) { return q.p[i].tiles_a * q.p[i].parts; };
    auto class_size = [&](int i) {
        int sz = 0;
        for (int j = 0; j < n; ++j)
            if (q.p[j].A == q.p[i].A) sz += size_of(j);
        return sz;
    };
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0; --j) {
            const int a = order[j], b = order[j - 1];
            const int ka = coarse ? class_size(a) : size_of(a), kb = coarse ? class_size(b) : size_of(b);
            const bool before = ka > kb || (ka == kb && coarse && q.p[a].A < q.p[b].A && q.p[a].A != q.p[b].A);
            if (!before) break;
            std::swap(order[j], order[j - 1]);
        }
    auto place = [&](int lo, int hi, int ta_lo, int ta_hi, int g) -> bool {  // problems order[lo..hi), narrow tiles [ta_lo, ta_hi)
        int sz = 0;
        for (int oi = lo; oi < hi; ++oi) sz += (std::min(ta_hi, q.p[order[oi]].tiles_a) - std::min(ta_lo, q.p[order[oi]].tiles_a)) * q.p[order[oi]].parts;
        int x = 0;
        for (int k = 1; k < 8; ++k)
            if (load[k] < load[x]) x = k;
        if (load[x] + sz > CAP) return false;
        for (int oi = lo; oi < hi; ++oi) {
            const SpTnProb& p = q.p[order[oi]];
            for (int ta = ta_lo; ta < ta_hi && ta < p.tiles_a; ++ta)
                for (int c = 0; c < p.parts; ++c) q.map[8 * (load[x]++) + x] = (uint32_t)(p.pp_lo + ta * p.parts + c) | ((uint32_t)g << 16);
        }
        return true;
    };
    for (int lo = 0; lo < n;) {
        int hi = lo + 1;
        if (coarse)
            while (hi < n && q.p[order[hi]].A == q.p[order[lo]].A) ++hi;
        int tmax = 0;
        for (int oi = lo; oi < hi; ++oi) tmax = std::max(tmax, q.p[order[oi]].tiles_a);
        for (int g = 0; g < q.G; ++g) {
            if (coarse) {
                if (!place(lo, hi, 0, tmax, g)) return 0;
            } else {
                for (int ta = 0; ta < tmax; ++ta)
                    if (!place(lo, hi, ta, ta + 1, g)) return 0;
            }
        }
        lo = hi;
    }
    max_load = 0;
    for (int k = 0; k < 8; ++k) max_load = std::max(max_load, load[k]);
    return (unsigned)(8 * max_load);
}
static unsigned sp_tn_map(SpTnParams& q) {
    // largest row-group count (from the one-round estimate down to 3 fewer) for which no XCD holds more than 32 workgroups (a
    // second residency round on one XCD doubles the launch); coarse units when they cost at most 5 % of the workgroups
    int mx = 0;
    const int G0 = q.G;
    int g_fine = 0, g_coarse = 0;
    for (int G = G0; G >= 1 && G >= G0 - 3 && g_fine == 0; --G) {
        q.G = G;
        if (sp_tn_map_units(q, false, mx) != 0 && mx <= 32) g_fine = G;
    }
    for (int G = G0; G >= 1 && G >= G0 - 3 && g_coarse == 0; --G) {
        q.G = G;
        if (sp_tn_map_units(q, true, mx) != 0 && mx <= 32) g_coarse = G;
    }
    if (g_coarse > 0 && g_coarse * 20 >= (g_fine > 0 ? g_fine : G0) * 19) {
        q.G = g_coarse;
        return sp_tn_map_units(q, true, mx);
    }
    q.G = g_fine > 0 ? g_fine : G0;
    return sp_tn_map_units(q, false, mx);
}
// Tune.tn (desc.sel_tn): 0 tiled k_tn only, 2 streaming whenever the shape allows, 1 (default): streaming when every wave gets
// >= 8 slabs (below that the launch is latency-bound and the tiled kernel's 64-row chunks win: measured on the stage-2 / 3 shapes)
static int64_t sp_tn_part_bytes(const mtlora_linear_desc* d, const Segs& sg) {
    const int nb = sp_tn_nb(d);
    if (nb == 0) return 0;
    int64_t n_pp = 0;
    for (int o = 0; o < sg.n; ++o)
        if (sg.rp[o] > 0) n_pp += mtl_ceil_div(sg.rp[o], 64) * (d->N / (32 * nb) + d->K / (32 * nb));
    const int64_t wgs = n_pp > SP_TN_WGS ? n_pp : SP_TN_WGS;
    return wgs * (2 * nb * 1024) * 4;
}
template <typename T>
static void launch_sp_tn(SpTnParams& q, int nb, hipStream_t s, double xb, double fl, const char* tag) {
    if constexpr (sizeof(T) == 2) {
        q.n_slabs = (int)mtl_ceil_div(q.M, 32);
        const unsigned grid = sp_tn_map(q);  // (q.G set by the caller; the caller checked that the table fits)
        {
            mtl_prof_tag("sp_tn %s np%d pp%d G%d nb%d", tag, q.n_prob, q.n_pp, q.G, nb);
            MtlProfScope prof(PK_TN, xb, s, xb, fl);
#define MTL_SP_TN(NBV, NSV)                                                                     \
    do {                                                                                        \
        constexpr size_t lds = (size_t)SP_WAVES * NSV * (SP_TN_NARROW + 32 * NBV * 64);         \
        MTL_RAISE_LDS((k_sp_tn<T, NBV, NSV>), SP_LDS_MAX);                                      \
        hipLaunchKernelGGL((k_sp_tn<T, NBV, NSV>), dim3(grid), dim3(64 * SP_WAVES), lds, s, q); \
    } while (0)
            if (nb == 4)
                MTL_SP_TN(4, 1);
            else
                MTL_SP_TN(3, 2);
#undef MTL_SP_TN
        }
        MtlProfScope prof(PK_REDUCE, 0.0, s);
        if (nb == 4)
            hipLaunchKernelGGL(k_sp_tn_reduce<4>, dim3(8, (unsigned)q.n_pp), dim3(256 * SP_TN_RG), 0, s, q);
        else
            hipLaunchKernelGGL(k_sp_tn_reduce<3>, dim3(6, (unsigned)q.n_pp), dim3(256 * SP_TN_RG), 0, s, q);
    }
}

static PackParams make_pack_params(const mtlora_linear_desc* d, const Segs& sg, const float* A_s, const float* B_s, const float* const* A_t,
                                   const float* const* B_t) {
    const float keep_scale = mtl_make_dropout(d->dropout_p, 0).enabled() ? 1.f / (1.f - d->dropout_p) : 1.f;
    PackParams pp;
    pp.s = sg;
    pp.K = (int)d->K;
    pp.N = (int)d->N;
    for (int o = 0; o < MAXO; ++o) {
        pp.A[o] = nullptr;
        pp.B[o] = nullptr;
        pp.alpha[o] = 0.f;
    }
    pp.A[0] = A_s;
    pp.B[0] = B_s;
    pp.alpha[0] = d->scale_s * keep_scale;
    for (int t = 0; t < d->T; ++t) {
        pp.A[t + 1] = A_t[t];
        pp.B[t + 1] = B_t[t];
        pp.alpha[t + 1] = d->scale_t[t] * (d->has_x_tasks ? 1.f : keep_scale);
    }
    return pp;
}
static unsigned pack_blocks(const mtlora_linear_desc* d, const Segs& sg) {
    const int64_t tr = mtl_ceil_div(sg.R, PK_T);
    int64_t blocks = tr * (mtl_ceil_div(d->K, PK_T) + mtl_ceil_div(d->N, PK_T));  // one 64 x 64 tile of A_cat / B_cat per workgroup
    if (blocks > 2048) blocks = 2048;
    return (unsigned)(blocks > 0 ? blocks : 1);
}
template <typename T>
static void launch_pack(const mtlora_linear_desc* d, const Segs& sg, const CtxLayout& L, unsigned char* pk, const float* A_s,
                        const float* B_s, const float* const* A_t, const float* const* B_t, hipStream_t s) {
    const PackParams pp = make_pack_params(d, sg, A_s, B_s, A_t, B_t);
    MtlProfScope prof(PK_PACK, 0.0, s);
    hipLaunchKernelGGL(k_pack<T>, dim3(pack_blocks(d, sg)), dim3(256), 0, s, pp, reinterpret_cast<T*>(pk + L.a_cat),
                       reinterpret_cast<T*>(pk + L.b_cat), reinterpret_cast<T*>(pk + L.at_cat), reinterpret_cast<T*>(pk + L.bt_cat),
                       reinterpret_cast<float*>(pk + L.alpha), reinterpret_cast<T*>(pk + L.a_proj),
                       reinterpret_cast<T*>(pk + L.bt_proj), reinterpret_cast<T*>(pk + L.b_frag), reinterpret_cast<T*>(pk + L.at_frag));
}

template <typename T>
static int fwd_impl(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* W,
                    const float* bias, const float* A_s, const float* B_s, const float* const* A_t,
                    const float* const* B_t, void* y_s, void* const* y_t, void* ctx, hipStream_t s, void* a_s = nullptr,
                    void* const* a_t = nullptr) {
    const Segs sg = make_segs(d);
    const Tune tu = make_tune(d);
    const CtxLayout L = ctx_layout(d, sg);
    unsigned char* c = reinterpret_cast<unsigned char*>(ctx);
    unsigned char* pk = d->packed ? const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(d->packed)) : c;
    T* a_cat = reinterpret_cast<T*>(pk + L.a_cat);
    T* b_cat = reinterpret_cast<T*>(pk + L.b_cat);
    T* at_cat = reinterpret_cast<T*>(pk + L.at_cat);
    T* bt_cat = reinterpret_cast<T*>(pk + L.bt_cat);
    float* alpha = reinterpret_cast<float*>(pk + L.alpha);
    (void)at_cat;
    (void)bt_cat;
    T* Pm = reinterpret_cast<T*>(c + L.p);
    const DropoutCfg dc = mtl_make_dropout(d->dropout_p, d->seed, d->seed_offset);

    if (sg.R > 0) {
        if (!d->packed) launch_pack<T>(d, sg, L, pk, A_s, B_s, A_t, B_t, s);

        // T = 0 layers with a short reduction: ONE wave-streaming launch (projection in registers, stream.h)
        if (d->T == 0 && d->mode == 0) {
            SpLinParams q = {};
            q.act = x;
            q.w = W;
            q.proj = pk + L.a_proj;
            q.expand = pk + L.b_frag;
            q.bias = bias;
            q.out = y_s;
            q.out2 = a_s;
            q.pout = Pm;
            q.ld_out = d->N;
            q.ldp = sg.R;
            q.M = d->M;
            q.n_cols = (int)d->N;
            q.R = sg.R;
            q.mask_act = 1;
            q.mask_lr = 0;
            q.drop = dc;
            SpXresPlan pl;
            const double b8d = (double)sizeof(T) * d->M * (d->K + (double)d->N);
            const double fl = 2.0 * d->M * (double)d->K * d->N + 2.0 * d->M * (double)sg.r[0] * (d->K + d->N);
            if (sp_xres_plan<T>(tu, q, (int)d->K, pl)) {
                launch_sp_xres<T>(q, pl, a_s != nullptr, s, PK_NT_FWD_MAIN, b8d + (a_s ? (double)sizeof(T) * d->M * d->N : 0.0), b8d, fl);
                return MTLORA_OK;
            }
            // a long reduction into few output columns (the Mlp's fc2 at stage 0: 384 -> 96): the accumulator-resident form, whose
            // projection sees the dropout-masked activation -- the P pass and its re-read of X disappear here too
            SpAresPlan pa;
            q.estep = q.estep2 = 0;
            if (a_s == nullptr && sp_ares_plan<T>(tu, q, (int)d->K, pa)) {
                launch_sp_ares<T>(q, pa, s, PK_NT_FWD_MAIN, b8d, b8d, fl);
                return MTLORA_OK;
            }
        }
        {
            // P = alpha * D(X) A^T  (per source)
            NtParams q = {};
            q.n_act = 1;
            q.ld_act = d->K;
            q.wgt = a_cat;
            q.ld_wgt = d->K;
            q.M = d->M;
            q.K = (int)d->K;
            q.alpha = alpha;
            q.n_out = 1;
            q.out[0].ptr = Pm;
            q.out[0].use_base = 1;
            q.ld_out = sg.R;
            q.drop = dc;
            if (d->T > 0 && d->has_x_tasks) {
                q.nz = 0;
                for (int o = 0; o < sg.n; ++o) {
                    if (sg.rp[o] == 0) continue;
                    if (o > 0 && (d->hid & MTLORA_HID_P_GIVEN)) continue;  // (task columns of P: mtlora_mlp_hid_proj wrote them)
                    q.zact[q.nz] = (o == 0) ? x : x_t[o - 1];
                    q.zrow0[q.nz] = sg.off[o];
                    q.zrows[q.nz] = sg.rp[o];
                    q.zmask[q.nz] = (o == 0) ? 1 : 0;
                    ++q.nz;
                }
                q.n_rows = 0;
            } else {
                q.act[0] = x;
                q.act_mask = 1;
                q.n_rows = sg.R;
                q.nz = 0;
            }
            {
                // (HID_P_GIVEN: the tasks' own inputs are read -- implicitly -- by k_hid_proj, which carries their 8(d) bytes)
                const double xb = (double)sizeof(T) * ((d->has_x_tasks && !(d->hid & MTLORA_HID_P_GIVEN)) ? d->T : 0) * d->M * d->K;
                double rsum = 0.0;  // un-padded ranks
                for (int o = 0; o < sg.n; ++o) rsum += sg.r[o];
                // wave-streaming form (stream.h) when the alpha-scaled factor rows fit in LDS next to the slab ring
                SpProjParams sp = {};
                sp.wproj = pk + L.a_proj;
                sp.out = Pm;
                sp.ld_out = sg.R;
                sp.M = d->M;
                sp.K = (int)d->K;
                sp.Rw = sg.R;
                sp.drop = dc;
                for (int o = 0; o < sg.n; ++o) {
                    const bool own = d->T > 0 && d->has_x_tasks;
                    if (!own && o > 0) break;
                    if (own && sg.rp[o] == 0) continue;
                    if (own && o > 0 && (d->hid & MTLORA_HID_P_GIVEN)) continue;
                    SpSrc& ss = sp.src[sp.n_src++];
                    ss.act = (o == 0) ? x : x_t[o - 1];
                    ss.col_lo = own ? sg.off[o] : 0;
                    ss.col_hi = own ? sg.off[o] + sg.rp[o] : sg.used;
                    ss.blk_lo = ss.col_lo / 32;
                    ss.n_blk = (ss.col_hi - 1) / 32 - ss.blk_lo + 1;
                    ss.mask = (o == 0) ? 1 : 0;
                }
                int ch = 0;
                const int ns = sp_proj_plan<T>(tu, sp, ch);
                PqParams pq = {};
                const int mb = pq_plan<T>(tu, sp, pq);
                // k_pq first: forced (3: the "[pq]" test family), instead of a one-slot k_sp_proj ring (which cannot overlap its loads), and
                // instead of k_sp_proj for launches of about one residency round (tools/pq_times.py: stage 2 of Swin-T 13 vs 16 us, Swin-B 14
                // vs 21 us); where k_sp_proj does not fit at all, k_sp_projk's single-round rule comes first (stage 3: 15 vs 21 us)
                if (sp.n_src == 0)
                    ;  // (no source left: r_s = 0 and the task columns are given)
                else if (mb > 0 && (tu.projk == 3 || ns == 1 || (ns > 1 && pq_one_round(tu, pq, mb))))
                    launch_pq<T>(pq, mb, s, PK_NT_FWD_P, xb, xb, 2.0 * d->M * d->K * rsum);
                else if (ns > 0)
                    launch_sp_proj<T>(tu, sp, ch, ns, s, PK_NT_FWD_P, xb, xb, 2.0 * d->M * d->K * rsum);
                else if (sp_projk_plan<T>(tu, sp, ch))
                    launch_sp_projk<T>(tu, sp, ch, s, PK_NT_FWD_P, xb, xb, 2.0 * d->M * d->K * rsum);
                else if (mb > 0)
                    launch_pq<T>(pq, mb, s, PK_NT_FWD_P, xb, xb, 2.0 * d->M * d->K * rsum);
                else
                    launch_nt<T>(tu, q, s, PK_NT_FWD_P, xb, xb, 2.0 * d->M * d->K * rsum);
            }
        }
    }

    // all outputs
    NtParams m = {};
    m.act[0] = x;
    m.n_act = 1;
    m.ld_act = d->K;
    m.wgt = W;
    m.ld_wgt = d->K;
    m.M = d->M;
    m.n_rows = (int)d->N;
    m.K = (int)d->K;
    m.bias = bias;
    m.L = Pm;
    m.ldL = sg.R;
    m.Rm = b_cat;
    m.ldR = sg.R;
    m.ld_out = d->N;
    m.drop = dc;
    m.n_out = 1 + d->T;
    for (int o = 0; o < sg.n; ++o) {
        NtOut& O = m.out[o];
        O.ptr = (o == 0) ? y_s : y_t[o - 1];
        O.seg_lo = sg.off[o];
        O.seg_hi = sg.off[o] + sg.rp[o];
        O.use_base = 1;
        O.mask_lr = 0;
        O.fold = (o == 0 && d->mode == 1 && d->T > 0) ? 1 : 0;
        O.act = (o == 0) ? a_s : (a_t ? a_t[o - 1] : nullptr);
    }
    if (d->hid & MTLORA_HID_FWD_BASE) {  // the Mlp's fc1 with implicit task hiddens: the shared output and the bare pretrained product
        m.n_out = 2;
        NtOut& O = m.out[1];
        O.ptr = const_cast<void*>(d->hid_ptr);
        O.seg_lo = O.seg_hi = 0;
        O.use_base = 1;
        O.mask_lr = 0;
        O.fold = 0;
        O.act = nullptr;
    }
    int n_actout = 0;  // GELU second outputs: one more M x N write each
    for (int o = 0; o < m.n_out; ++o) n_actout += m.out[o].act ? 1 : 0;
    {
        const double b8d = (double)sizeof(T) * d->M * (d->K + (double)(1 + d->T) * d->N);  // (SURVEY 8(d) counts every module output)
        double rsum = 0.0;
        for (int o = 0; o < sg.n; ++o) rsum += sg.r[o];
        const double fl = 2.0 * d->M * d->K * d->N + 2.0 * d->M * d->N * rsum;
        const bool plain = sg.R == 0;
        launch_nt<T>(tu, m, s, plain ? PK_NT_PLAIN_FWD : PK_NT_FWD_MAIN, b8d + (double)sizeof(T) * d->M * (double)n_actout * d->N,
                     plain ? 0.0 : b8d, fl);
    }
    return MTLORA_OK;
}

constexpr int64_t TN_TARGET_CTAS = 512;
struct BwdScratch {
    int64_t q, g, part, total;
    int nsplit;
    int64_t rows_per_split;
};
static BwdScratch bwd_scratch(const mtlora_linear_desc* d, const Segs& sg) {
    const int es = mtl_elem_size(d->dtype);
    BwdScratch S;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        int64_t at = o;
        o += mtl_round_up(bytes, 256);
        return at;
    };
    S.q = take(d->M * sg.R * es);
    S.g = take(d->T > 0 ? d->M * d->N * es : 0);  // G = sum of the output gradients (matrixv2 factors; pre-summed dX operand)
    // TN tiles (rank side x wide side), dB and dA per output
    int64_t tiles = 0;
    for (int oo = 0; oo < sg.n; ++oo) {
        if (sg.rp[oo] == 0) continue;
        tiles += mtl_ceil_div(sg.rp[oo], TN_A) * mtl_ceil_div(d->N, TN_B);  // dB (as P^T dY)
        tiles += mtl_ceil_div(sg.rp[oo], TN_A) * mtl_ceil_div(d->K, TN_B);  // dA
    }
    // <= 512 workgroups = 2 per CU, all resident; >= 256 rows per split so that the fp32 partial tiles stay a small part of the traffic
    int nsplit = 1;
    if (tiles > 0) {
        nsplit = (int)(TN_TARGET_CTAS / tiles);  // floor: the whole grid is resident at once (no second round)
        const int64_t max_by_rows = mtl_ceil_div(d->M, 256);
        if (nsplit > max_by_rows) nsplit = (int)max_by_rows;
        if (nsplit > 256) nsplit = 256;
        if (nsplit < 1) nsplit = 1;
    }
    S.nsplit = nsplit;
    int64_t rps = mtl_ceil_div(d->M > 0 ? d->M : 1, nsplit);
    rps = mtl_round_up(rps, 64);
    S.rows_per_split = rps;
    {
        const int64_t tiled = tiles * nsplit * (int64_t)TN_TILE * 4, streamed = sp_tn_part_bytes(d, sg);
        S.part = take(tiled > streamed ? tiled : streamed);
    }
    S.total = o;
    return S;
}

template <typename T>
static int bwd_impl(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* Wt,
                    const void* dy_s, const void* const* dy_t, const void* ctx, void* dx, void* const* dx_t,
                    float* dA_s, float* dB_s, float* const* dA_t, float* const* dB_t, void* scratch, hipStream_t s,
                    const void* gate_s = nullptr, const void* const* gate_t = nullptr) {
    const Segs sg = make_segs(d);
    const Tune tu = make_tune(d);
    const CtxLayout L = ctx_layout(d, sg);
    const BwdScratch S = bwd_scratch(d, sg);
    const unsigned char* c = reinterpret_cast<const unsigned char*>(ctx);
    const unsigned char* pk = d->packed ? reinterpret_cast<const unsigned char*>(d->packed) : c;
    const T* at_cat = reinterpret_cast<const T*>(pk + L.at_cat);
    const T* bt_cat = reinterpret_cast<const T*>(pk + L.bt_cat);
    const float* alpha = reinterpret_cast<const float*>(pk + L.alpha);
    const T* Pm = reinterpret_cast<const T*>(c + L.p);
    unsigned char* sc = reinterpret_cast<unsigned char*>(scratch);
    T* Qm = reinterpret_cast<T*>(sc + S.q);
    T* Gm = reinterpret_cast<T*>(sc + S.g);
    float* part = reinterpret_cast<float*>(sc + S.part);
    const DropoutCfg dc = mtl_make_dropout(d->dropout_p, d->seed, d->seed_offset);
    const bool v2 = d->mode == 1 && d->T > 0;
    const bool hid_q = (d->hid & MTLORA_HID_Q_GIVEN) != 0;  // fc1 of an Mlp with implicit task hiddens (hid.hip): Q task columns given

    // gradient sources per output
    const void* dy[MAXO];
    dy[0] = dy_s;
    for (int t = 0; t < d->T; ++t) dy[t + 1] = dy_t ? dy_t[t] : nullptr;
    int n_dy = 0;
    const void* dy_all[MAXO];
    for (int o = 0; o < sg.n; ++o)
        if (dy[o]) dy_all[n_dy++] = dy[o];

    // G = sum of every output gradient, materialised when
    //  * matrixv2: the shared factors see G, or
    //  * the dX kernel would otherwise re-sum the n_dy sources once per output n-tile (it re-reads every dY panel for
    //    each of the ceil(K/128) n-tiles: 3..24x for fc2): one k_sum pass + the single-source kernel moves
    //    (n_dy + 1 + n_tiles) MN bytes instead of n_dy * n_tiles * MN.
    const void* dy_shared = dy[0];
    const bool do_dx = d->bwd_phase != 2, do_factors = d->bwd_phase != 1;  // (phase 2 re-derives the same operand table)
    bool presum = n_dy > 1 && dx && mtl_ceil_div(d->K, TILE) >= 3;
    bool have_g = false;
    // layers with task outputs ('matrix' mode, every output has a gradient): ONE wave-streaming pass over the 1 + T gradient
    // tensors forms Q (all segments) AND G = sum_o dY_o (k_sp_projsum, stream.h); the dX launch then reads G alone
    bool q_done = false;
    if constexpr (sizeof(T) == 2) {
        if (d->T >= 1 && d->mode == 0 && do_dx && dx && n_dy == 1 + d->T && sg.rp[0] > 0 && sg.rp[0] <= 64) {
            SpProjParams sp = {};
            sp.wproj = pk + L.bt_proj;
            sp.out = Qm;
            sp.ld_out = sg.R;
            sp.M = d->M;
            sp.K = (int)d->N;
            sp.Rw = sg.R;
            sp.drop = dc;
            bool ok = true;
            for (int o = 0; o < sg.n; ++o) {
                if (sg.rp[o] == 0 || (o > 0 && sg.rp[o] > 32)) ok = false;
                SpSrc& ss = sp.src[sp.n_src++];
                ss.act = dy[o];
                ss.col_lo = sg.off[o];
                ss.col_hi = sg.off[o] + sg.rp[o];
                ss.blk_lo = ss.col_lo / 32;
                ss.n_blk = (ss.col_hi - 1) / 32 - ss.blk_lo + 1;
                ss.mask = 0;
                if (o > 0 && ss.n_blk != 1) ok = false;
            }
            if (ok && sp.src[sp.n_src - 1].blk_lo - sp.src[1].blk_lo + 1 > SP_PS_MAXT) ok = false;  // task segments span too many blocks
            int ch = 0;
            const int ns = ok ? sp_proj_plan<T>(tu, sp, ch) : 0;
            if (ns > 0 && d->M * d->N * 2 < ((int64_t)1 << 32) - 64) {
                double rsum = 0.0;
                for (int o = 0; o < sg.n; ++o) rsum += sg.r[o];
                launch_sp_projsum<T>(tu, sp, ch, ns, Gm, s, PK_NT_BWD_Q, 0.0, 0.0, 2.0 * d->M * d->N * rsum);
                q_done = true;
                presum = true;
                have_g = true;
            }
        }
    }
    if ((v2 || presum) && n_dy > 1) {
        have_g = true;
    }
    if (have_g && do_dx && !q_done) {
        SumParams sp;
        sp.n = n_dy;
        for (int i = 0; i < n_dy; ++i) sp.src[i] = dy_all[i];
        sp.nvec = d->M * d->N / ET<T>::VEC;
        int64_t blocks = mtl_ceil_div(sp.nvec, 256);
        if (blocks > 4096) blocks = 4096;
        if (blocks > 0) {
            MtlProfScope prof(PK_SUM, (double)sizeof(T) * d->M * d->N * (n_dy + 1), s);
            hipLaunchKernelGGL(k_sum<T>, dim3((unsigned)blocks), dim3(256), 0, s, sp, Gm);
        }
    }
    if (v2 && n_dy > 0) dy_shared = have_g ? (const void*)Gm : dy_all[0];
    const void* dyo[MAXO];  // gradient feeding factor o
    for (int o = 0; o < sg.n; ++o) dyo[o] = (o == 0) ? dy_shared : dy[o];

    // T = 0 layers whose input is narrow: Q, the masked rank part and dY W in ONE wave-streaming pass over dY (stream.h)
    bool sp_dx_done = false;
    if (d->T == 0 && d->mode == 0 && do_dx && dx && dyo[0] && sg.R > 0) {
        SpLinParams q = {};
        q.act = dyo[0];
        q.w = Wt;
        q.proj = pk + L.bt_proj;
        q.expand = pk + L.at_frag;
        q.out = dx;
        q.pout = Qm;
        q.ld_out = d->K;
        q.ldp = sg.R;
        q.M = d->M;
        q.n_cols = (int)d->K;
        q.R = sg.R;
        q.mask_act = 0;
        q.mask_lr = 1;
        q.drop = dc;
        const double b8d = (double)sizeof(T) * d->M * ((double)d->N + d->K);
        const double fl = 2.0 * d->M * (double)d->N * d->K + 2.0 * d->M * (double)sg.r[0] * (d->K + d->N);
        SpAresPlan pl;
        SpXresPlan px;
        if (!gate_s && sp_ares_plan<T>(tu, q, (int)d->N, pl)) {
            launch_sp_ares<T>(q, pl, s, PK_NT_BWD_DX, b8d, b8d, fl);
            sp_dx_done = true;
        } else {
            // wide input, short reduction (the Mlp's fc2: dX has 4 C columns, the reduction C <= 192): the activation-resident form,
            // with the GELU' gate of the fused Mlp in its epilogue
            q.gate = gate_s;
            if (sp_xres_plan<T>(tu, q, (int)d->N, px)) {
                launch_sp_xres<T>(q, px, false, s, PK_NT_BWD_DX, b8d + (gate_s ? (double)sizeof(T) * d->M * d->K : 0.0), b8d, fl, gate_s != nullptr);
                sp_dx_done = true;
            }
            q.gate = nullptr;
        }
    }

    // Q[:, seg_o] = alpha_o * dY_o B_o   (zero where the output got no gradient)
    if (sg.R > 0 && do_dx && !sp_dx_done && !q_done) {
        bool any_missing = false;
        for (int o = 0; o < sg.n; ++o)
            if (sg.rp[o] > 0 && !dyo[o]) any_missing = true;
        if (any_missing && !hid_q) mtl_zero_async(Qm, (size_t)(d->M * sg.R * sizeof(T)), s);  // (hid_q: the task columns are given)
        NtParams q = {};
        q.n_act = 1;
        q.ld_act = d->N;
        q.wgt = bt_cat;
        q.ld_wgt = d->N;
        q.M = d->M;
        q.K = (int)d->N;
        q.alpha = alpha;
        q.n_out = 1;
        q.out[0].ptr = Qm;
        q.out[0].use_base = 1;
        q.ld_out = sg.R;
        q.drop = dc;
        q.nz = 0;
        for (int o = 0; o < sg.n; ++o) {
            if (sg.rp[o] == 0 || !dyo[o]) continue;
            q.zact[q.nz] = dyo[o];
            q.zrow0[q.nz] = sg.off[o];
            q.zrows[q.nz] = sg.rp[o];
            q.zmask[q.nz] = 0;
            ++q.nz;
        }
        if (q.nz > 0) {
            double rsum = 0.0;
            for (int o = 0; o < sg.n; ++o)
                if (sg.rp[o] > 0 && dyo[o]) rsum += sg.r[o];
            SpProjParams sp = {};
            sp.wproj = pk + L.bt_proj;
            sp.out = Qm;
            sp.ld_out = sg.R;
            sp.M = d->M;
            sp.K = (int)d->N;
            sp.Rw = sg.R;
            sp.drop = dc;
            for (int o = 0; o < sg.n; ++o) {
                if (sg.rp[o] == 0 || !dyo[o]) continue;
                SpSrc& ss = sp.src[sp.n_src++];
                ss.act = dyo[o];
                ss.col_lo = sg.off[o];
                ss.col_hi = sg.off[o] + sg.rp[o];
                ss.blk_lo = ss.col_lo / 32;
                ss.n_blk = (ss.col_hi - 1) / 32 - ss.blk_lo + 1;
                ss.mask = 0;
            }
            int ch = 0;
            const int ns = sp_proj_plan<T>(tu, sp, ch);
            PqParams pq = {};
            const int mb = pq_plan<T>(tu, sp, pq);
            if (mb > 0 && (tu.projk == 3 || ns == 1 || (ns > 1 && pq_one_round(tu, pq, mb))))
                launch_pq<T>(pq, mb, s, PK_NT_BWD_Q, 0.0, 0.0, 2.0 * d->M * d->N * rsum);
            else if (ns > 0)
                launch_sp_proj<T>(tu, sp, ch, ns, s, PK_NT_BWD_Q, 0.0, 0.0, 2.0 * d->M * d->N * rsum);
            else if (sp_projk_plan<T>(tu, sp, ch))
                launch_sp_projk<T>(tu, sp, ch, s, PK_NT_BWD_Q, 0.0, 0.0, 2.0 * d->M * d->N * rsum);
            else if (mb > 0)
                launch_pq<T>(pq, mb, s, PK_NT_BWD_Q, 0.0, 0.0, 2.0 * d->M * d->N * rsum);
            else
                launch_nt<T>(tu, q, s, PK_NT_BWD_Q, 0.0, 0.0, 2.0 * d->M * d->N * rsum);
        }
    }

    // dX = G W + keep .* (Q_s A_s [+ sum_t Q_t A_t]),  dX_t = Q_t A_t
    if (do_dx && !sp_dx_done) {
        NtParams m = {};
        RankOutParams rank_out = {};
        int rank_out_rp = 8;
        bool rank_out_gated = false;
        if (presum && have_g) {
            m.n_act = 1;
            m.act[0] = Gm;
        } else {
            m.n_act = n_dy;
            for (int i = 0; i < n_dy; ++i) m.act[i] = dy_all[i];
        }
        if (hid_q && d->hid_ptr) {  // G = dH_s + sum_t dH_t was formed by k_hid_bwd
            m.n_act = 1;
            m.act[0] = d->hid_ptr;
        }
        m.ld_act = d->N;
        m.wgt = Wt;
        m.ld_wgt = d->N;
        m.M = d->M;
        m.n_rows = (int)d->K;
        m.K = (n_dy > 0 || (hid_q && d->hid_ptr)) ? (int)d->N : 0;
        m.L = Qm;
        m.ldL = sg.R;
        m.Rm = at_cat;
        m.ldR = sg.R;
        m.ld_out = d->K;
        m.drop = dc;
        m.n_out = 1;
        NtOut& O = m.out[0];
        O.ptr = dx;
        O.use_base = 1;
        O.mask_lr = 1;
        O.gate = gate_s;
        if (d->T > 0 && d->has_x_tasks) {
            O.seg_lo = sg.off[0];
            O.seg_hi = sg.off[0] + sg.rp[0];
            // small task ranks: the task outputs are a streaming elementwise kernel of their own (k_rank_out), not tile passes
            if constexpr (sizeof(T) == 2) {
                bool ok = tu.sp != 0 && dx_t != nullptr && d->K % 8 == 0 && d->K / 8 <= 256 && d->M > 0;
                int rpm = 0, gates = 0, outs = 0;
                for (int t = 0; t < d->T; ++t) {
                    if (!dx_t || !dx_t[t]) continue;
                    ++outs;
                    rpm = sg.rp[t + 1] > rpm ? sg.rp[t + 1] : rpm;
                    gates += (gate_t && gate_t[t]) ? 1 : 0;
                    ok = ok && sg.rp[t + 1] > 0 && !misaligned(dx_t[t]) && !(gate_t && gate_t[t] && misaligned(gate_t[t]));
                }
                for (int t = 0; t < d->T; ++t)
                    if (dx_t && dx_t[t]) ok = ok && sg.rp[t + 1] == rpm;  // one register geometry per launch
                ok = ok && outs > 0 && rpm <= 16 && (gates == 0 || gates == outs) && !misaligned(Qm) && (sg.R % 8) == 0;
                if (ok) {
                    rank_out.Q = Qm;
                    rank_out.Acat = pk + L.a_cat;
                    rank_out.ldq = sg.R;
                    rank_out.M = d->M;
                    rank_out.K = (int)d->K;
                    for (int t = 0; t < d->T; ++t) {
                        if (!dx_t[t]) continue;
                        const int i = rank_out.n_t++;
                        rank_out.seg[i] = sg.off[t + 1];
                        rank_out.rp[i] = sg.rp[t + 1];
                        rank_out.out[i] = dx_t[t];
                        rank_out.gate[i] = gate_t ? gate_t[t] : nullptr;
                    }
                    rank_out_rp = rpm <= 8 ? 8 : 16;
                    rank_out_gated = gates > 0;
                }
            }
            for (int t = 0; t < d->T && rank_out.n_t == 0; ++t) {
                if (!dx_t || !dx_t[t]) continue;
                NtOut& Ot = m.out[m.n_out++];
                Ot.ptr = dx_t[t];
                Ot.seg_lo = sg.off[t + 1];
                Ot.seg_hi = sg.off[t + 1] + sg.rp[t + 1];
                Ot.use_base = 0;
                Ot.mask_lr = 0;
                Ot.fold = 0;
                Ot.gate = gate_t ? gate_t[t] : nullptr;
            }
        } else {
            O.seg_lo = 0;
            O.seg_hi = sg.used;  // (not R: the Q pass writes segments only, the pad columns of Q hold whatever the scratch held)
        }
        if (dx) {
            int n_gate = 0;  // the fused GELU backward reads the pre-activation of every gated output (algorithmic: gelu'(h) needs h)
            for (int o = 0; o < m.n_out; ++o) n_gate += m.out[o].gate ? 1 : 0;
            const int n_xt = rank_out.n_t > 0 ? 0 : (d->has_x_tasks ? d->T : 0);  // task outputs written by THIS launch
            const double b8d = (double)sizeof(T) * d->M * ((double)n_dy * d->N + (double)(1 + n_xt) * d->K);
            double rsum = 0.0;
            for (int o = 0; o < sg.n; ++o)
                if (sg.rp[o] > 0 && dyo[o]) rsum += sg.r[o];
            const double fl = (n_dy > 0 ? 2.0 * d->M * d->N * d->K : 0.0) + 2.0 * d->M * d->K * rsum;
            const bool plain = sg.R == 0;
            launch_nt<T>(tu, m, s, plain ? PK_NT_PLAIN_DX : PK_NT_BWD_DX, b8d + (double)sizeof(T) * d->M * (double)n_gate * d->K,
                         plain ? 0.0 : b8d, fl);
        }
        if constexpr (sizeof(T) == 2) {
            if (rank_out.n_t > 0) {
                const double ob = (double)sizeof(T) * d->M * (double)rank_out.n_t * d->K;
                mtl_prof_tag("rank_out M%lld K%lld nt%d rp%d gate%d", (long long)d->M, (long long)d->K, rank_out.n_t, rank_out_rp, rank_out_gated ? 1 : 0);
                MtlProfScope prof(PK_NT_BWD_DX, ob * (rank_out_gated ? 2.0 : 1.0), s, ob, 0.0);
                const int nchunk = (int)(d->K / 8), rpb = 256 / nchunk;
                int64_t bx = mtl_ceil_div(d->M, (int64_t)rpb * 4);
                const int64_t cap = (int64_t)num_cu(tu) * 8 / rank_out.n_t > 0 ? (int64_t)num_cu(tu) * 8 / rank_out.n_t : 1;
                if (bx > cap) bx = cap;
                const dim3 g((unsigned)bx, (unsigned)rank_out.n_t);
                if (rank_out_gated && rank_out_rp == 8)
                    hipLaunchKernelGGL((k_rank_out<T, true, 8>), g, dim3(256), 0, s, rank_out);
                else if (rank_out_gated)
                    hipLaunchKernelGGL((k_rank_out<T, true, 16>), g, dim3(256), 0, s, rank_out);
                else if (rank_out_rp == 8)
                    hipLaunchKernelGGL((k_rank_out<T, false, 8>), g, dim3(256), 0, s, rank_out);
                else
                    hipLaunchKernelGGL((k_rank_out<T, false, 16>), g, dim3(256), 0, s, rank_out);
            }
        }
    }

    // dA_o = Q_o^T D(X_o),  dB_o = dY_o^T P_o
    if (sg.R > 0 && do_factors) {
        TnParams tp = {};
        tp.M = d->M;
        tp.nsplit = S.nsplit;
        tp.rows_per_split = S.rows_per_split;
        tp.drop = dc;
        float* pp = part;
        int max_tiles = 0;
        for (int o = 0; o < sg.n; ++o) {
            const bool q_given = hid_q && o > 0;  // Q[:, seg_o] came from k_hid_bwd (which also returns dB_o): dA_o only
            if (sg.rp[o] == 0 || (!dyo[o] && !q_given)) continue;
            float* dAo = (o == 0) ? dA_s : (dA_t ? dA_t[o - 1] : nullptr);
            float* dBo = (o == 0) ? dB_s : (dB_t ? dB_t[o - 1] : nullptr);
            if (!dyo[o]) dBo = nullptr;
            if (dBo) {  // (N x r_o) = dY_o^T P[:, seg_o], evaluated as its transpose P[:, seg_o]^T dY_o
                TnProblem& p = tp.p[tp.n_prob++];
                p.A = Pm;
                p.lda = sg.R;
                p.a0 = sg.off[o];
                p.Na = sg.rp[o];
                p.B = dyo[o];
                p.ldb = d->N;
                p.b0 = 0;
                p.Nb = (int)d->N;
                p.b_mask = 0;
                p.tiles_a = (int)mtl_ceil_div(p.Na, TN_A);
                p.tiles_b = (int)mtl_ceil_div(p.Nb, TN_B);
                p.part = pp;
                pp += (int64_t)p.tiles_a * p.tiles_b * S.nsplit * TN_TILE;
                p.out = dBo;
                p.out_a = sg.r[o];
                p.out_b = (int)d->N;
                p.ldo = sg.r[o];
                p.transpose = 1;
                if (p.tiles_a * p.tiles_b > max_tiles) max_tiles = p.tiles_a * p.tiles_b;
            }
            if (dAo) {  // (r_o x K) = Q[:, seg_o]^T . D(X_o)
                TnProblem& p = tp.p[tp.n_prob++];
                p.A = Qm;
                p.lda = sg.R;
                p.a0 = sg.off[o];
                p.Na = sg.rp[o];
                const bool own_x = (o > 0 && d->has_x_tasks);
                p.B = own_x ? x_t[o - 1] : x;
                p.ldb = d->K;
                p.b0 = 0;
                p.Nb = (int)d->K;
                p.b_mask = own_x ? 0 : 1;
                p.tiles_a = (int)mtl_ceil_div(p.Na, TN_A);
                p.tiles_b = (int)mtl_ceil_div(p.Nb, TN_B);
                p.part = pp;
                pp += (int64_t)p.tiles_a * p.tiles_b * S.nsplit * TN_TILE;
                p.out = dAo;
                p.out_a = sg.r[o];
                p.out_b = (int)d->K;
                p.ldo = (int)d->K;
                p.transpose = 0;
                if (p.tiles_a * p.tiles_b > max_tiles) max_tiles = p.tiles_a * p.tiles_b;
            }
        }
        bool tn_done = false;
        if constexpr (sizeof(T) == 2) {
            const int nb = (tu.sp != 0 && tu.tn != 0) ? sp_tn_nb(d) : 0;
            bool ok = nb != 0 && tp.n_prob > 0;
            SpTnParams q = {};
            for (int i = 0; ok && i < tp.n_prob; ++i) {
                const TnProblem& p = tp.p[i];
                SpTnProb& r = q.p[i];
                ok = ok && p.b0 == 0 && p.Nb == p.ldb && p.Nb % (32 * nb) == 0 && !misaligned(p.A) && !misaligned(p.B) && (p.lda % 8) == 0 &&
                     (p.a0 % 8) == 0;
                r.A = p.A;
                r.B = p.B;
                r.lda = p.lda;
                r.ldb = p.ldb;
                r.a0 = p.a0;
                r.Na = p.Na;
                r.Nb = p.Nb;
                r.b_mask = p.b_mask;
                r.tiles_a = (int)mtl_ceil_div(p.Na, 64);
                r.parts = p.Nb / (32 * nb);
                r.pp_lo = q.n_pp;
                r.transpose = p.transpose;
                r.out = p.out;
                r.out_a = p.out_a;
                r.out_b = p.out_b;
                r.ldo = p.ldo;
                q.n_pp += r.tiles_a * r.parts;
            }
            if (ok) {
                const int mode = tu.tn;
                q.n_prob = tp.n_prob;
                q.G = sp_tn_groups(tu, q.n_pp, d->M);
                ok = q.n_pp <= SP_TN_WGS && q.G < 65536 && (mode == 2 || (mode == 1 && mtl_ceil_div(d->M, 32) >= (int64_t)8 * q.G * SP_WAVES));
                if (ok) {
                    SpTnParams probe = q;
                    ok = sp_tn_map(probe) != 0;
                    q.G = probe.G;
                }
            }
            if (ok) {
                q.n_prob = tp.n_prob;
                q.M = d->M;
                q.part = part;
                q.drop = dc;
                const double xb = (double)sizeof(T) * d->M * (double)(1 + (d->has_x_tasks ? d->T : 0)) * d->K;
                double fl = 0.0;
                for (int i = 0; i < tp.n_prob; ++i) fl += 2.0 * d->M * (double)tp.p[i].out_a * tp.p[i].out_b;
                char tag[96];
                snprintf(tag, sizeof(tag), "M%lld K%lld N%lld T%d", (long long)d->M, (long long)d->K, (long long)d->N, d->T);
                launch_sp_tn<T>(q, nb, s, xb, fl, tag);
                tn_done = true;
            }
        }
        if (!tn_done && tp.n_prob > 0 && d->M > 0) {
            {
                mtl_prof_tag("M%lld K%lld N%lld T%d np%d ns%d tiles%d", (long long)d->M, (long long)d->K, (long long)d->N, d->T, tp.n_prob,
                             S.nsplit, max_tiles);
                const double xb = (double)sizeof(T) * d->M * (double)(1 + (d->has_x_tasks ? d->T : 0)) * d->K;
                double fl = 0.0;
                for (int i = 0; i < tp.n_prob; ++i) fl += 2.0 * d->M * (double)tp.p[i].out_a * tp.p[i].out_b;
                MtlProfScope prof(PK_TN, xb, s, xb, fl);
                hipLaunchKernelGGL(k_tn<T>, dim3((unsigned)S.nsplit, (unsigned)max_tiles, (unsigned)tp.n_prob),
                                   dim3(256), 0, s, tp);
            }
            MtlProfScope prof(PK_REDUCE, 0.0, s);
            hipLaunchKernelGGL(k_tn_reduce, dim3(TN_TILE / 1024, (unsigned)max_tiles, (unsigned)tp.n_prob), dim3(256 * TN_RG), 0, s, tp);
        }
    }
    return MTLORA_OK;
}

// ---- Mlp with implicit task hidden tensors (hid.hip): shape rules, launch geometry, the two launches
struct HidPlan {
    int tg, rr, nthr, nw, groups, n_wg;
};
static int hid_check(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    int st = check_desc(d1);
    if (st != MTLORA_OK) return st;
    st = check_desc(d2);
    if (st != MTLORA_OK) return st;
    if (d1->dtype == MTLORA_F32 || d1->dtype != d2->dtype) return MTLORA_ERR_UNSUPPORTED;
    if (d1->M != d2->M || d1->N != d2->K || d1->T != d2->T || d1->T < 1) return MTLORA_ERR_UNSUPPORTED;
    if (d1->mode != 0 || d2->mode != 0 || !d1->has_x_tasks || !d2->has_x_tasks) return MTLORA_ERR_UNSUPPORTED;
    if (d1->N % 128 != 0 || d1->M <= 0) return MTLORA_ERR_UNSUPPORTED;
    int rmax = 0;
    for (int t = 0; t < d1->T; ++t) {
        if (d1->r_t[t] < 1 || d1->r_t[t] > 8 || d2->r_t[t] < 1 || d2->r_t[t] > 8) return MTLORA_ERR_UNSUPPORTED;
        rmax = std::max(rmax, std::max(d1->r_t[t], d2->r_t[t]));
    }
    // the VALU forms hold a whole row per workgroup (two columns per thread: <= 2048 columns); the MFMA forms take any number of chunks
    const bool chunked = d1->sel_stream != 1 && rmax <= 4;  // (N % 128 == 0: a chunk width exists in both directions)
    if (d1->N > 2048 && !chunked) return MTLORA_ERR_UNSUPPORTED;
    return MTLORA_OK;
}
static HidPlan hid_plan(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    HidPlan pl;
    int rmax = 0;
    for (int t = 0; t < d1->T; ++t) rmax = std::max(rmax, std::max(d1->r_t[t], d2->r_t[t]));
    pl.rr = rmax <= 4 ? 4 : 8;
    pl.nthr = (int)(d1->N / 2);
    pl.nw = pl.nthr / 64;
    // register budget: the backward kernel holds 8 * TG * RR factor + accumulator registers per lane: TG * RR = 16 needs the 256-register
    // budget of <= 512-thread workgroups (2 waves per SIMD), 8 fits the 168 registers of 768 threads, 1024 threads are capped at 128
    const int cap = pl.nthr <= 512 ? 16 : 8;
    pl.tg = std::max(1, cap / pl.rr);
    if (pl.nthr > 768 && pl.rr == 8) pl.tg = 1;
    pl.groups = (d1->T + pl.tg - 1) / pl.tg;
    const Tune tu = make_tune(d1);
    const int wg_per_cu = std::max(1, (8 + pl.nw - 1) / pl.nw);
    const int64_t nblk = (d1->M + HID_RB - 1) / HID_RB;
    pl.n_wg = (int)std::min<int64_t>(nblk, (int64_t)num_cu(tu) * wg_per_cu);
    return pl;
}
static int64_t hid_part_bytes(const mtlora_linear_desc* d1, const HidPlan& pl) {
    // (either form of the backward kernel: VALU n_wg x tg tasks, MFMA <= one workgroup per CU x HID_TG tasks)
    const Tune tu = make_tune(d1);
    const int64_t slots = std::max<int64_t>((int64_t)pl.n_wg * pl.tg, (int64_t)num_cu(tu) * HID_TG);
    return slots * 2 * pl.rr * d1->N * 4 + 256;
}

static void hid_launch_valu(int dtype, bool bwd, const HidPlan& pl, const HidParams& q, hipStream_t s) {
    HidLaunch L = {};
    L.kind = bwd ? 1 : 0;
    L.dtype = dtype;
    L.tg = pl.tg;
    L.rr = pl.rr;
    L.nthr = pl.nthr;
    L.n_wg = pl.n_wg;
    mtli_hid_launch(&L, &q, s);
}
// MFMA forms (hid.hip, k_hid_fwd_d / k_hid_bwd_d): rank <= 4, hidden width a multiple of 384 or 256 columns (one chunk per grid y)
struct HidDPlan {
    bool on;
    int hc, n_chunk, n_wg, groups;
    size_t lds_f, lds_b;
    int64_t rowpart_bytes;
};
static HidDPlan hid_d_plan(const mtlora_linear_desc* d1, const HidPlan& pl, bool bwd) {
    HidDPlan dp = {};
    const Tune tu = make_tune(d1);
    const int H = (int)d1->N;
    dp.hc = hid_d_chunk(H, bwd);
    dp.on = tu.sp != 0 && pl.rr == 4 && dp.hc != 0;
    if (!dp.on) return dp;
    dp.n_chunk = H / dp.hc;
    dp.groups = (d1->T + HID_TG - 1) / HID_TG;
    dp.lds_f = hid_d_lds_bytes(dp.hc, false);
    dp.lds_b = hid_d_lds_bytes(dp.hc, true);
    const int64_t nblk = (d1->M + 31) / 32;
    // workgroups per CU: as many as fit by LDS and by 12 waves of 168 registers
    const int per_cu = std::max(1, std::min((int)((size_t)(150 * 1024) / (bwd ? dp.lds_b : dp.lds_f)), (bwd ? 12 : 16) / (dp.hc / 32)));
    dp.n_wg = (int)std::min<int64_t>(nblk, std::max<int64_t>(1, (int64_t)num_cu(tu) * per_cu / dp.n_chunk));
    dp.rowpart_bytes = (int64_t)dp.n_chunk * d1->M * 16 * 4 + 256;
    return dp;
}
static void hid_launch_d(int dtype, bool bwd, const HidDPlan& dp, const HidParams& q, hipStream_t s) {
    HidLaunch L = {};
    L.kind = bwd ? 3 : 2;
    L.dtype = dtype;
    L.rr = 4;
    L.hc = dp.hc;
    L.n_chunk = dp.n_chunk;
    L.n_wg = dp.n_wg;
    L.nthr = dp.hc * 2;
    L.lds = bwd ? dp.lds_b : dp.lds_f;
    mtli_hid_launch(&L, &q, s);
}
static int64_t hid_fwd_scratch_bytes(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    const HidPlan pl = hid_plan(d1, d2);
    const HidDPlan dp = hid_d_plan(d1, pl, false);
    return dp.on ? dp.rowpart_bytes : 256;
}
static int64_t hid_bwd_part_bytes(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    const HidPlan pl = hid_plan(d1, d2);
    const HidDPlan dp = hid_d_plan(d1, pl, true);
    if (!dp.on) return hid_part_bytes(d1, pl);
    // [chunk][n_wg][nt <= 4][2][4][hc] factor-gradient partials, then the row-sum partials
    return (int64_t)dp.n_chunk * dp.n_wg * HID_TG * 2 * 4 * dp.hc * 4 + 256 + dp.rowpart_bytes;
}

template <typename T>
static int hid_proj_impl(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2, const void* h_base, const void* ctx1, void* ctx2,
                         void* scratch, hipStream_t s) {
    const Segs s1 = make_segs(d1), s2 = make_segs(d2);
    const CtxLayout L1 = ctx_layout(d1, s1), L2 = ctx_layout(d2, s2);
    const unsigned char* c1 = reinterpret_cast<const unsigned char*>(ctx1);
    unsigned char* c2 = reinterpret_cast<unsigned char*>(ctx2);
    const unsigned char* pk1 = d1->packed ? reinterpret_cast<const unsigned char*>(d1->packed) : c1;
    const unsigned char* pk2 = reinterpret_cast<const unsigned char*>(d2->packed);
    const HidPlan pl = hid_plan(d1, d2);
    const HidDPlan dp = hid_d_plan(d1, pl, false);
    const int H = (int)d1->N;
    const int tg = dp.on ? HID_TG : pl.tg;
    const int groups = (d1->T + tg - 1) / tg;
    for (int gI = 0; gI < groups; ++gI) {
        HidParams q = {};
        q.hbase = h_base;
        q.p1 = c1 + L1.p;
        q.p2 = c2 + L2.p;
        q.b1t = pk1 + L1.bt_cat;
        q.a2 = pk2 + L2.a_cat;
        q.alpha1 = reinterpret_cast<const float*>(pk1 + L1.alpha);
        q.alpha2 = reinterpret_cast<const float*>(pk2 + L2.alpha);
        q.rowpart = reinterpret_cast<float*>(scratch);
        q.M = d1->M;
        q.H = H;
        q.ldp1 = s1.R;
        q.ldp2 = s2.R;
        q.nt = std::min(tg, d1->T - gI * tg);
        for (int i = 0; i < HID_TG; ++i) {  // (slots past nt: valid offsets, never stored -- hid.hip)
            const int t = gI * tg + (i < q.nt ? i : 0);
            q.off1[i] = s1.off[1 + t];
            q.off2[i] = s2.off[1 + t];
        }
        const double hb = (double)sizeof(T) * d1->M * H;
        mtl_prof_tag("hid_fwd%s M%lld H%d T%d nt%d rr%d", dp.on ? "_d" : "", (long long)d1->M, H, d1->T, q.nt, pl.rr);
        {
            MtlProfScope prof(PK_NT_FWD_P, hb, s, (double)sizeof(T) * d1->M * (double)q.nt * H, 0.0);
            if (dp.on)
                hid_launch_d(d1->dtype, false, dp, q, s);
            else
                hid_launch_valu(d1->dtype, false, pl, q, s);
        }
        if (dp.on) {  // (its own profiler record: one record per dispatch, tools/pmc_traffic.py zips the two lists)
            mtl_prof_tag("hid_rows_finish M%lld chunks%d", (long long)d1->M, dp.n_chunk);
            MtlProfScope prof(PK_REDUCE, 0.0, s);
            mtli_hid_rows_finish(d1->dtype, q.rowpart, dp.n_chunk, d1->M, q.nt, q.alpha2, q.off2, q.p2, q.ldp2, s);
        }
    }
    return MTLORA_OK;
}

template <typename T>
static int hid_bwd_impl(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2, const void* h_base, const void* dh_s, const void* ctx1,
                        const void* ctx2, const void* scratch2, void* scratch1, void* g, float* const* dB1_t, float* const* dA2_t, void* part,
                        hipStream_t s) {
    const Segs s1 = make_segs(d1), s2 = make_segs(d2);
    const CtxLayout L1 = ctx_layout(d1, s1), L2 = ctx_layout(d2, s2);
    const BwdScratch S1 = bwd_scratch(d1, s1), S2 = bwd_scratch(d2, s2);
    const unsigned char* c1 = reinterpret_cast<const unsigned char*>(ctx1);
    const unsigned char* c2 = reinterpret_cast<const unsigned char*>(ctx2);
    const unsigned char* pk1 = d1->packed ? reinterpret_cast<const unsigned char*>(d1->packed) : c1;
    const unsigned char* pk2 = d2->packed ? reinterpret_cast<const unsigned char*>(d2->packed) : c2;
    const HidPlan pl = hid_plan(d1, d2);
    const HidDPlan dp = hid_d_plan(d1, pl, true);
    const int H = (int)d1->N;
    const int tg = dp.on ? HID_TG : pl.tg;
    const int groups = (d1->T + tg - 1) / tg;
    const int rr = pl.rr;
    const int64_t fact_bytes = dp.on ? (int64_t)dp.n_chunk * dp.n_wg * HID_TG * 2 * 4 * dp.hc * 4 + 256 : 0;
    for (int gI = 0; gI < groups; ++gI) {
        HidParams q = {};
        q.hbase = h_base;
        q.p1 = c1 + L1.p;
        q.q2 = reinterpret_cast<const unsigned char*>(scratch2) + S2.q;
        q.q1 = reinterpret_cast<unsigned char*>(scratch1) + S1.q;
        q.gsrc = gI == 0 ? dh_s : g;
        q.g = g;
        q.b1t = pk1 + L1.bt_cat;
        q.a2 = pk2 + L2.a_cat;
        q.alpha1 = reinterpret_cast<const float*>(pk1 + L1.alpha);
        q.alpha2 = reinterpret_cast<const float*>(pk2 + L2.alpha);
        q.part = reinterpret_cast<float*>(part);
        q.rowpart = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(part) + fact_bytes);
        q.M = d1->M;
        q.H = H;
        q.ldp1 = s1.R;
        q.ldq1 = s1.R;
        q.ldq2 = s2.R;
        q.nt = std::min(tg, d1->T - gI * tg);
        for (int i = 0; i < HID_TG; ++i) {  // (slots past nt: valid offsets, never stored -- hid.hip)
            const int t = gI * tg + (i < q.nt ? i : 0);
            q.off1[i] = s1.off[1 + t];
            q.off2[i] = s2.off[1 + t];
        }
        HidRedParams r = {};
        for (int i = 0; i < q.nt; ++i) {
            const int t = gI * tg + i;
            r.r[i] = 0;
            r.dB1[i] = dB1_t ? dB1_t[t] : nullptr;
            r.dA2[i] = dA2_t ? dA2_t[t] : nullptr;
        }
        {
            mtl_prof_tag("hid_bwd%s M%lld H%d T%d nt%d rr%d", dp.on ? "_d" : "", (long long)d1->M, H, d1->T, q.nt, rr);
            const double hb = (double)sizeof(T) * d1->M * H;
            MtlProfScope prof(PK_NT_BWD_DX, 3.0 * hb, s, (double)sizeof(T) * d1->M * (double)q.nt * H, 0.0);
            if (dp.on)
                hid_launch_d(d1->dtype, true, dp, q, s);
            else
                hid_launch_valu(d1->dtype, true, pl, q, s);
        }
        if (dp.on) {
            mtl_prof_tag("hid_rows_finish M%lld chunks%d", (long long)d1->M, dp.n_chunk);
            MtlProfScope prof(PK_REDUCE, 0.0, s);
            mtli_hid_rows_finish(d1->dtype, q.rowpart, dp.n_chunk, d1->M, q.nt, q.alpha1, q.off1, q.q1, q.ldq1, s);
        }
        // dB1_t (fc1's N x r_t, un-padded rank d1->r_t) and dA2_t (fc2's r_t x K) share one reduce: per kind the un-padded rank differs
        // only if the two layers were built with different task ranks -- reduce them separately then
        r.part = q.part;
        r.n_wg = dp.on ? dp.n_wg : pl.n_wg;
        r.nt = q.nt;
        r.RR = rr;
        r.H = H;
        r.chunk_cols = dp.on ? dp.hc : H;
        const int64_t per = (int64_t)q.nt * 2 * rr * H;
        bool same = true;
        for (int i = 0; i < q.nt; ++i) same = same && d1->r_t[gI * tg + i] == d2->r_t[gI * tg + i];
        MtlProfScope prof(PK_REDUCE, 0.0, s);
        if (same) {
            for (int i = 0; i < q.nt; ++i) r.r[i] = d1->r_t[gI * tg + i];
            mtli_hid_reduce(&r, per, s);
        } else {
            HidRedParams rb = r, ra = r;
            for (int i = 0; i < q.nt; ++i) {
                rb.r[i] = d1->r_t[gI * tg + i];
                rb.dA2[i] = nullptr;
                ra.r[i] = d2->r_t[gI * tg + i];
                ra.dB1[i] = nullptr;
            }
            mtli_hid_reduce(&rb, per, s);
            MtlProfScope prof2(PK_REDUCE, 0.0, s);
            mtli_hid_reduce(&ra, per, s);
        }
    }
    return MTLORA_OK;
}

}  // namespace

extern "C" {

int mtlora_mlp_hid_supported(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    return (d1 && d2 && hid_check(d1, d2) == MTLORA_OK) ? 1 : 0;
}

int64_t mtlora_mlp_hid_fwd_scratch_bytes(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    if (!d1 || !d2 || hid_check(d1, d2) != MTLORA_OK) return -1;
    return hid_fwd_scratch_bytes(d1, d2);
}

int64_t mtlora_mlp_hid_bwd_scratch_bytes(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2) {
    if (!d1 || !d2 || hid_check(d1, d2) != MTLORA_OK) return -1;
    return hid_bwd_part_bytes(d1, d2);
}

int mtlora_mlp_hid_proj(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2, const void* h_base, const void* ctx1, void* ctx2,
                        void* scratch, int64_t scratch_bytes, void* stream) {
    if (!d1 || !d2) return MTLORA_ERR_NULL;
    const int st = hid_check(d1, d2);
    if (st != MTLORA_OK) return st;
    if (!h_base || !ctx1 || !ctx2 || !d2->packed || !scratch) return MTLORA_ERR_NULL;
    if (misaligned(h_base) || misaligned(ctx1) || misaligned(ctx2) || misaligned(d2->packed) || misaligned(d1->packed) || misaligned(scratch))
        return MTLORA_ERR_ALIGN;
    if (scratch_bytes < hid_fwd_scratch_bytes(d1, d2)) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int r = d1->dtype == MTLORA_F16 ? hid_proj_impl<f16>(d1, d2, h_base, ctx1, ctx2, scratch, s)
                                          : hid_proj_impl<bf16>(d1, d2, h_base, ctx1, ctx2, scratch, s);
    if (r != MTLORA_OK) return r;
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_mlp_hid_bwd(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2, const void* h_base, const void* dh_s, const void* ctx1,
                       const void* ctx2, const void* scratch2, void* scratch1, void* g, float* const* dB1_t, float* const* dA2_t, void* part,
                       int64_t part_bytes, void* stream) {
    if (!d1 || !d2) return MTLORA_ERR_NULL;
    const int st = hid_check(d1, d2);
    if (st != MTLORA_OK) return st;
    if (!h_base || !ctx1 || !ctx2 || !scratch1 || !scratch2 || !g || !part) return MTLORA_ERR_NULL;
    if (misaligned(h_base) || misaligned(dh_s) || misaligned(ctx1) || misaligned(ctx2) || misaligned(scratch1) || misaligned(scratch2) ||
        misaligned(g) || misaligned(part))
        return MTLORA_ERR_ALIGN;
    if (part_bytes < hid_bwd_part_bytes(d1, d2)) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int r = d1->dtype == MTLORA_F16 ? hid_bwd_impl<f16>(d1, d2, h_base, dh_s, ctx1, ctx2, scratch2, scratch1, g, dB1_t, dA2_t, part, s)
                                          : hid_bwd_impl<bf16>(d1, d2, h_base, dh_s, ctx1, ctx2, scratch2, scratch1, g, dB1_t, dA2_t, part, s);
    if (r != MTLORA_OK) return r;
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int64_t mtlora_linear_ctx_bytes(const mtlora_linear_desc* d) {
    if (check_desc(d) != MTLORA_OK) return -1;
    const Segs sg = make_segs(d);
    return ctx_layout(d, sg).total + 256;
}

int64_t mtlora_linear_bwd_scratch_bytes(const mtlora_linear_desc* d) {
    if (check_desc(d) != MTLORA_OK) return -1;
    const Segs sg = make_segs(d);
    return bwd_scratch(d, sg).total + 256;
}

int64_t mtlora_linear_packed_bytes(const mtlora_linear_desc* d) {
    if (check_desc(d) != MTLORA_OK) return -1;
    const Segs sg = make_segs(d);
    return ctx_layout(d, sg).pack_total + 256;
}

static int pack_check(const mtlora_linear_desc* d, const float* A_s, const float* B_s, const float* const* A_t, const float* const* B_t,
                      const void* packed, int64_t packed_bytes) {
    const int st = check_desc(d);
    if (st != MTLORA_OK) return st;
    if (!packed) return MTLORA_ERR_NULL;
    if (misaligned(packed)) return MTLORA_ERR_ALIGN;
    if (d->r_s > 0 && (!A_s || !B_s)) return MTLORA_ERR_NULL;
    for (int t = 0; t < d->T; ++t)
        if (!A_t || !B_t || !A_t[t] || !B_t[t]) return MTLORA_ERR_NULL;
    const Segs sg = make_segs(d);
    if (packed_bytes < ctx_layout(d, sg).pack_total) return MTLORA_ERR_WORKSPACE;
    return MTLORA_OK;
}

int mtlora_linear_pack(const mtlora_linear_desc* d, const float* A_s, const float* B_s, const float* const* A_t,
                       const float* const* B_t, void* packed, int64_t packed_bytes, void* stream) {
    const int st = pack_check(d, A_s, B_s, A_t, B_t, packed, packed_bytes);
    if (st != MTLORA_OK) return st;
    const Segs sg = make_segs(d);
    if (sg.R == 0) return MTLORA_OK;
    const CtxLayout L = ctx_layout(d, sg);
    unsigned char* pk = reinterpret_cast<unsigned char*>(packed);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == MTLORA_F32)
        launch_pack<float>(d, sg, L, pk, A_s, B_s, A_t, B_t, s);
    else if (d->dtype == MTLORA_F16)
        launch_pack<f16>(d, sg, L, pk, A_s, B_s, A_t, B_t, s);
    else
        launch_pack<bf16>(d, sg, L, pk, A_s, B_s, A_t, B_t, s);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int64_t mtlora_linear_pack_entry_bytes(void) { return (int64_t)sizeof(PackEntry); }

int mtlora_linear_pack_entry(const mtlora_linear_desc* d, const float* A_s, const float* B_s, const float* const* A_t,
                             const float* const* B_t, void* packed, int64_t packed_bytes, void* entry_host) {
    const int st = pack_check(d, A_s, B_s, A_t, B_t, packed, packed_bytes);
    if (st != MTLORA_OK) return st;
    if (!entry_host) return MTLORA_ERR_NULL;
    const Segs sg = make_segs(d);
    if (sg.R == 0) return MTLORA_ERR_SHAPE;  // nothing to pack: such a layer does not belong in a table
    const CtxLayout L = ctx_layout(d, sg);
    PackEntry e;
    memset(&e, 0, sizeof(e));
    e.pp = make_pack_params(d, sg, A_s, B_s, A_t, B_t);
    e.dst = reinterpret_cast<unsigned char*>(packed);
    const int64_t off[9] = {L.a_cat, L.b_cat, L.at_cat, L.bt_cat, L.alpha, L.a_proj, L.bt_proj, L.b_frag, L.at_frag};
    for (int i = 0; i < 9; ++i) e.off[i] = off[i];
    memcpy(entry_host, &e, sizeof(e));
    return MTLORA_OK;
}

int mtlora_linear_pack_table(const void* table_dev, int n_entries, int dtype, void* stream) {
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16 && dtype != MTLORA_F16) return MTLORA_ERR_DTYPE;
    if (n_entries < 0 || n_entries > 65535) return MTLORA_ERR_SHAPE;
    if (n_entries == 0) return MTLORA_OK;
    if (!table_dev) return MTLORA_ERR_NULL;
    if (((uintptr_t)table_dev & 7u) != 0) return MTLORA_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const PackEntry* tb = reinterpret_cast<const PackEntry*>(table_dev);
    MtlProfScope prof(PK_PACK, 0.0, s);
    const dim3 g(256, (unsigned)n_entries);  // up to 256 workgroups per layer walk its 20 - 800 tiles (the others leave at once)
    if (dtype == MTLORA_F32)
        hipLaunchKernelGGL(k_pack_table<float>, g, dim3(256), 0, s, tb);
    else if (dtype == MTLORA_F16)
        hipLaunchKernelGGL(k_pack_table<f16>, g, dim3(256), 0, s, tb);
    else
        hipLaunchKernelGGL(k_pack_table<bf16>, g, dim3(256), 0, s, tb);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

static int linear_fwd_entry(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* W,
                            const float* bias, const float* A_s, const float* B_s, const float* const* A_t,
                            const float* const* B_t, void* y_s, void* const* y_t, void* a_s, void* const* a_t, void* ctx,
                            int64_t ctx_bytes, void* stream) {
    int st = check_desc(d);
    if (st != MTLORA_OK) return st;
    if (!x || !W || !y_s) return MTLORA_ERR_NULL;
    if (d->r_s > 0 && !d->packed && (!A_s || !B_s)) return MTLORA_ERR_NULL;
    if (misaligned(x) || misaligned(W) || misaligned(y_s) || misaligned(bias) || misaligned(d->packed)) return MTLORA_ERR_ALIGN;
    const bool hid_base = (d->hid & MTLORA_HID_FWD_BASE) != 0, hid_p = (d->hid & MTLORA_HID_P_GIVEN) != 0;
    if (hid_base && (!d->hid_ptr || d->T < 1 || !d->has_x_tasks || d->mode != 0)) return MTLORA_ERR_NULL;
    if (hid_base && misaligned(d->hid_ptr)) return MTLORA_ERR_ALIGN;
    if (hid_p && (d->T < 1 || !d->has_x_tasks || d->mode != 0)) return MTLORA_ERR_UNSUPPORTED;
    for (int t = 0; t < d->T; ++t) {
        if (!hid_base && (!y_t || !y_t[t])) return MTLORA_ERR_NULL;  // (HID_FWD_BASE: no task outputs)
        if (!d->packed && (!A_t || !B_t || !A_t[t] || !B_t[t])) return MTLORA_ERR_NULL;
        if (!hid_base && misaligned(y_t[t])) return MTLORA_ERR_ALIGN;
        if (d->has_x_tasks && !hid_p && (!x_t || !x_t[t])) return MTLORA_ERR_NULL;  // (HID_P_GIVEN: x_t is not read)
        if (d->has_x_tasks && !hid_p && misaligned(x_t[t])) return MTLORA_ERR_ALIGN;
    }
    const Segs sg = make_segs(d);
    if (sg.R > 0) {
        if (!ctx) return MTLORA_ERR_NULL;
        if (misaligned(ctx)) return MTLORA_ERR_ALIGN;
        if (ctx_bytes < ctx_layout(d, sg).total) return MTLORA_ERR_WORKSPACE;
    }
    if (d->M == 0) return MTLORA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == MTLORA_F32)
        st = fwd_impl<float>(d, x, x_t, W, bias, A_s, B_s, A_t, B_t, y_s, y_t, ctx, s, a_s, a_t);
    else if (d->dtype == MTLORA_F16)
        st = fwd_impl<f16>(d, x, x_t, W, bias, A_s, B_s, A_t, B_t, y_s, y_t, ctx, s, a_s, a_t);
    else
        st = fwd_impl<bf16>(d, x, x_t, W, bias, A_s, B_s, A_t, B_t, y_s, y_t, ctx, s, a_s, a_t);
    if (st != MTLORA_OK) return st;
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_linear_fwd(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* W,
                      const float* bias, const float* A_s, const float* B_s, const float* const* A_t,
                      const float* const* B_t, void* y_s, void* const* y_t, void* ctx, int64_t ctx_bytes,
                      void* stream) {
    return linear_fwd_entry(d, x, x_t, W, bias, A_s, B_s, A_t, B_t, y_s, y_t, nullptr, nullptr, ctx, ctx_bytes, stream);
}

int mtlora_linear_fwd_gelu(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* W,
                           const float* bias, const float* A_s, const float* B_s, const float* const* A_t,
                           const float* const* B_t, void* y_s, void* const* y_t, void* a_s, void* const* a_t, void* ctx,
                           int64_t ctx_bytes, void* stream) {
    if (!a_s || !d) return MTLORA_ERR_NULL;
    if (misaligned(a_s)) return MTLORA_ERR_ALIGN;
    for (int t = 0; t < d->T && !(d->hid & MTLORA_HID_FWD_BASE); ++t) {  // (HID_FWD_BASE: no task outputs)
        if (!a_t || !a_t[t]) return MTLORA_ERR_NULL;
        if (misaligned(a_t[t])) return MTLORA_ERR_ALIGN;
    }
    return linear_fwd_entry(d, x, x_t, W, bias, A_s, B_s, A_t, B_t, y_s, y_t, a_s, a_t, ctx, ctx_bytes, stream);
}

static int linear_bwd_entry(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* Wt,
                            const void* dy_s, const void* const* dy_t, const void* ctx, int64_t ctx_bytes, void* dx,
                            void* const* dx_t, float* dA_s, float* dB_s, float* const* dA_t, float* const* dB_t,
                            void* scratch, int64_t scratch_bytes, const void* gate_s, const void* const* gate_t,
                            void* stream) {
    int st = check_desc(d);
    if (st != MTLORA_OK) return st;
    if (!x || !Wt) return MTLORA_ERR_NULL;
    if (misaligned(x) || misaligned(Wt) || misaligned(dx) || misaligned(dy_s)) return MTLORA_ERR_ALIGN;
    const Segs sg = make_segs(d);
    if (sg.R > 0) {
        if (!ctx || !scratch) return MTLORA_ERR_NULL;
        if (misaligned(ctx) || misaligned(scratch)) return MTLORA_ERR_ALIGN;
        if (ctx_bytes < ctx_layout(d, sg).total) return MTLORA_ERR_WORKSPACE;
        if (scratch_bytes < bwd_scratch(d, sg).total) return MTLORA_ERR_WORKSPACE;
    }
    if ((d->hid & MTLORA_HID_Q_GIVEN) && (d->T < 1 || !d->has_x_tasks || d->mode != 0 || misaligned(d->hid_ptr))) return MTLORA_ERR_UNSUPPORTED;
    for (int t = 0; t < d->T; ++t) {
        // a task's own input is only read for its factor gradient dA_t (an Mlp with implicit task hiddens asks fc2 for none: hid.hip)
        if (d->has_x_tasks && dA_t && dA_t[t] && (!x_t || !x_t[t])) return MTLORA_ERR_NULL;
        if (d->has_x_tasks && x_t && misaligned(x_t[t])) return MTLORA_ERR_ALIGN;
        if (dy_t && misaligned(dy_t[t])) return MTLORA_ERR_ALIGN;
        if (dx_t && misaligned(dx_t[t])) return MTLORA_ERR_ALIGN;
    }
    if (d->M == 0) return MTLORA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == MTLORA_F32)
        st = bwd_impl<float>(d, x, x_t, Wt, dy_s, dy_t, ctx, dx, dx_t, dA_s, dB_s, dA_t, dB_t, scratch, s, gate_s, gate_t);
    else if (d->dtype == MTLORA_F16)
        st = bwd_impl<f16>(d, x, x_t, Wt, dy_s, dy_t, ctx, dx, dx_t, dA_s, dB_s, dA_t, dB_t, scratch, s, gate_s, gate_t);
    else
        st = bwd_impl<bf16>(d, x, x_t, Wt, dy_s, dy_t, ctx, dx, dx_t, dA_s, dB_s, dA_t, dB_t, scratch, s, gate_s, gate_t);
    if (st != MTLORA_OK) return st;
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_linear_bwd(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* Wt,
                      const void* dy_s, const void* const* dy_t, const void* ctx, int64_t ctx_bytes, void* dx,
                      void* const* dx_t, float* dA_s, float* dB_s, float* const* dA_t, float* const* dB_t,
                      void* scratch, int64_t scratch_bytes, void* stream) {
    return linear_bwd_entry(d, x, x_t, Wt, dy_s, dy_t, ctx, ctx_bytes, dx, dx_t, dA_s, dB_s, dA_t, dB_t, scratch, scratch_bytes,
                            nullptr, nullptr, stream);
}

int mtlora_linear_bwd_gelu(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* Wt,
                           const void* dy_s, const void* const* dy_t, const void* ctx, int64_t ctx_bytes, void* dx,
                           void* const* dx_t, float* dA_s, float* dB_s, float* const* dA_t, float* const* dB_t,
                           void* scratch, int64_t scratch_bytes, const void* h_s, const void* const* h_t, void* stream) {
    if (!h_s || !dx) return MTLORA_ERR_NULL;
    if (misaligned(h_s)) return MTLORA_ERR_ALIGN;
    for (int t = 0; t < d->T && d->has_x_tasks; ++t) {
        if (dx_t && dx_t[t] && (!h_t || !h_t[t])) return MTLORA_ERR_NULL;
        if (h_t && misaligned(h_t[t])) return MTLORA_ERR_ALIGN;
    }
    return linear_bwd_entry(d, x, x_t, Wt, dy_s, dy_t, ctx, ctx_bytes, dx, dx_t, dA_s, dB_s, dA_t, dB_t, scratch, scratch_bytes,
                            h_s, h_t, stream);
}

// ---- out (Na x Nb, fp32) = a^T b reduced over the M rows: the weight gradient dW = dY^T X of a plain linear layer whose
// output is narrow (the decoder heads' last 1x1 convolutions, seg_hrnet.py:518-526: Na = classes, Nb = 1080, M = B*H*W).
// Same split-M TN kernel as the LoRA factor gradients (k_tn + fixed-order k_tn_reduce: deterministic).
static int tn_plan(int64_t M, int Na, int Nb, int& tiles_a, int& tiles_b, int& nsplit, int64_t& rps) {
    tiles_a = (int)mtl_ceil_div(Na, TN_A);
    tiles_b = (int)mtl_ceil_div(Nb, TN_B);
    const int64_t tiles = (int64_t)tiles_a * tiles_b;
    nsplit = (int)(TN_TARGET_CTAS / (tiles > 0 ? tiles : 1));
    const int64_t max_by_rows = mtl_ceil_div(M, 256);
    if (nsplit > max_by_rows) nsplit = (int)max_by_rows;
    if (nsplit > 256) nsplit = 256;
    if (nsplit < 1) nsplit = 1;
    rps = mtl_round_up(mtl_ceil_div(M > 0 ? M : 1, nsplit), 64);
    return MTLORA_OK;
}

int64_t mtlora_gemm_tn_scratch_bytes(int64_t M, int Na, int Nb) {
    if (M < 0 || Na <= 0 || Nb <= 0) return -1;
    int ta, tb, ns;
    int64_t rps;
    tn_plan(M, Na, Nb, ta, tb, ns, rps);
    return (int64_t)ta * tb * ns * TN_TILE * 4 + 256;
}

int mtlora_gemm_tn(const void* a, const void* b, float* out, int64_t M, int Na, int Nb, int64_t lda, int64_t ldb,
                   int dtype, void* scratch, int64_t scratch_bytes, void* stream) {
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16 && dtype != MTLORA_F16) return MTLORA_ERR_DTYPE;
    const int vec = dtype == MTLORA_F32 ? 4 : 8;
    if (M < 0 || M >= ((int64_t)1 << 31) || Na <= 0 || Nb <= 0 || lda < Na || ldb < Nb) return MTLORA_ERR_SHAPE;
    if (Na % vec || Nb % vec || lda % vec || ldb % vec) return MTLORA_ERR_ALIGN;
    if (Na > 1024) return MTLORA_ERR_UNSUPPORTED;  // narrow side only (every a-tile re-reads b)
    if (!out || !scratch || (M > 0 && (!a || !b))) return MTLORA_ERR_NULL;
    if (misaligned(a) || misaligned(b) || misaligned(scratch) || ((uintptr_t)out & 3u)) return MTLORA_ERR_ALIGN;
    if (scratch_bytes < mtlora_gemm_tn_scratch_bytes(M, Na, Nb) - 256) return MTLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        mtl_zero_async(out, (size_t)Na * Nb * 4, s);
        return MTLORA_OK;
    }
    TnParams tp = {};
    tp.M = M;
    int ta, tb;
    tn_plan(M, Na, Nb, ta, tb, tp.nsplit, tp.rows_per_split);
    tp.drop = mtl_make_dropout(0.f, 0, nullptr);
    tp.n_prob = 1;
    TnProblem& p = tp.p[0];
    p.A = a;
    p.B = b;
    p.lda = lda;
    p.ldb = ldb;
    p.a0 = 0;
    p.Na = Na;
    p.b0 = 0;
    p.Nb = Nb;
    p.b_mask = 0;
    p.tiles_a = ta;
    p.tiles_b = tb;
    p.part = reinterpret_cast<float*>(scratch);
    p.out = out;
    p.out_a = Na;
    p.out_b = Nb;
    p.ldo = Nb;
    p.transpose = 0;
    const int es = mtl_elem_size(dtype);
    {
        mtl_prof_tag("M%lld Na%d Nb%d", (long long)M, Na, Nb);
        MtlProfScope prof(PK_TN_PLAIN, (double)es * M * ((double)ta * Nb + Na), s, 0.0, 2.0 * M * (double)Na * Nb);
        if (dtype == MTLORA_F32)
            hipLaunchKernelGGL(k_tn<float>, dim3((unsigned)tp.nsplit, (unsigned)(ta * tb), 1), dim3(256), 0, s, tp);
        else if (dtype == MTLORA_F16)
            hipLaunchKernelGGL(k_tn<f16>, dim3((unsigned)tp.nsplit, (unsigned)(ta * tb), 1), dim3(256), 0, s, tp);
        else
            hipLaunchKernelGGL(k_tn<bf16>, dim3((unsigned)tp.nsplit, (unsigned)(ta * tb), 1), dim3(256), 0, s, tp);
    }
    MtlProfScope prof(PK_REDUCE, 0.0, s);
    hipLaunchKernelGGL(k_tn_reduce, dim3(TN_TILE / 1024, (unsigned)(ta * tb), 1), dim3(256 * TN_RG), 0, s, tp);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}
}
