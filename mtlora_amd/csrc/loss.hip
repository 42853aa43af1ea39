// loss.hip -- the loss end of the train step (SURVEY 8 a13 callers: models/swin_mtl.py:245 final F.interpolate +
// mtl_loss_schemes.py), fused:   loss_t( interpolate(low, scale S, bilinear, align_corners=False), label_t )
// and d loss_t / d low in ONE kernel per task.  The (B, C, S*h, S*w) upsampled logits, their fp32 copy, the
// log-softmax / probabilities and the three gradient tensors of the same size that the ATen sequence writes and
// re-reads (7 passes over 540 MB for the 21-class head at B = 32, 448 x 448) never exist: the kernel reads the
// low-resolution logits (34 MB) and the labels (26 MB) and writes the low-resolution gradient.
//
// One WAVE (a 64-thread workgroup) per tile of TQ x TR LOW-resolution pixels, TQ = 64 / S - 1 (7 at the models' S = 8): the
// output pixels whose bilinear taps include a pixel of the tile span (TQ + 1) S <= 64 columns -- one per lane -- and
// (TR + 1) S rows, which the wave walks top to bottom.  Per output row a lane rebuilds its pixel's C interpolated logits
// from the staged (TR + 2) x (TQ + 2) low-resolution neighbourhood (fp32 in LDS; the x-interpolated pair of a source cell is
// kept in registers across the S rows of the cell), evaluates the per-pixel loss gradient g[c] = d loss / d up ONCE, and
// applies the transpose of the interpolation as a GATHER in two steps:
//   along y  in registers: a0[c] += (1 - fy) g[c], a1[c] += fy g[c] belong to the low-res rows i0 and i0 + 1 of the current
//            source cell; when the walk enters the next cell, row i0 is complete
//   along x  through a [64][C] LDS image of a0: lane (qx, c) sums its 2S columns with the table weights and stores d low
// Fixed summation order, no float atomics: the result is deterministic.  The rim of half a cell around a tile belongs to two
// tiles (1.31 x 1.25 redundant evaluation at TQ = 7, TR = 4); the loss VALUE counts a pixel in the tile that owns it
// (oy / S, ox / S).  No workgroup barrier anywhere; the label of the next row is fetched while the current one is evaluated.
// (Rounds 2-5 gathered per LOW-resolution pixel -- 4 threads per pixel walking its 2S x 2S outputs and rebuilding logits +
// softmax for each: every output pixel evaluated 4 x: 452 / 236 / 173 / 163 us for the four heads of c2.  A first
// output-owned version with 256-thread workgroups, bands of rows and both gathers through LDS behind barriers ran at the
// same 460 us for the 21-class head: a third of the issue slots, 2 waves per SIMD waiting at barriers.)
// Index / weight arithmetic is PyTorch's (area_pixel_compute_source_index, align_corners=False:
// src = max((dst + 0.5) * in/out - 0.5, 0), i1 = i0 + (i0 < in - 1)), evaluated per output pixel and per (q, k) table
// entry, so the clamped borders need no special case.  S <= 32.
//
//   kind 0  SoftMaxwithLoss        (mtl_loss_schemes.py:22-39)  cross entropy, ignore_index, mean over valid pixels
//   kind 1  NormalsLoss            (:162-220, normalize=True, L1, size_average)
//   kind 2  BalancedCrossEntropy   (:42-89, size_average)       stat[0] = w = mean(1 - labels)
// Label-only statistics (valid count, mask sum, w) come from the caller (they do not depend on the prediction).
#include "common.h"

namespace {

struct UpLossParams {
    const void* low;     // (B, h, w, C)  channels-last low-resolution prediction
    const float* label;  // kinds 0, 2: (B, 1, H, W);  kind 1: (B, C, H, W)
    const float* stat;   // device scalars, see above
    void* dlow;          // (B, h, w, C)  d loss / d low  (same dtype as low)
    float* part;         // [gridDim.x]   per-workgroup share of the (already normalised) loss value
    int B, h, w, C, S;
    float ignore;
};

// low-res rows per tile: more rows = less rim (the (TR + 1) / TR redundancy), fewer rows = more waves to hide the walk's
// latencies; at B = 32, 56 x 56 -> 448 x 448 the kernel times are flat over TR = 2 .. 8 (more waves and more rim cancel), at B = 8
// TR = 2 is 1.8 x faster than TR = 8, at B = 64 6 % slower
constexpr int UL_TR = 4;

// per-output-pixel loss gradient g[c] = d loss / d up[c] (normalised), returns the pixel's share of the loss value;
// lab: the pixel's label (kinds 0, 2) or its C label channels (kind 1)
template <int KIND, int CMAX>
__device__ __forceinline__ float up_pixel(float (&up)[CMAX], float (&g)[CMAX], const float (&lab)[KIND == 1 ? CMAX : 1], int C,
                                          float ignore, float norm, float wneg) {
    float loss = 0.f;
    if (KIND == 0) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) g[c] = 0.f;
        if (lab[0] != ignore) {
            const int cls = (int)lab[0];
            float m = -3.0e38f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) m = c < C ? fmaxf(m, up[c]) : m;
            float sum = 0.f, ucls = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                up[c] = c < C ? __expf(up[c] - m) : 0.f;
                sum += up[c];
            }
            const float inv = 1.f / sum;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const float pc = up[c] * inv;
                ucls = c == cls ? pc : ucls;
                g[c] = norm * (pc - (c == cls ? 1.f : 0.f));
            }
            loss = -__logf(ucls) * norm;
        }
    } else if (KIND == 1) {
        float mk[CMAX];
        float r2 = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            mk[c] = (c < C && lab[c < (KIND == 1 ? CMAX : 1) ? c : 0] != ignore) ? 1.f : 0.f;
            r2 += up[c] * up[c];
        }
        const float r = sqrtf(r2), n = r + 1e-12f;
        float gc[CMAX], dot = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const float d = up[c] / n - (c < C ? lab[c < (KIND == 1 ? CMAX : 1) ? c : 0] : 0.f);
            gc[c] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * mk[c] * norm;
            dot += gc[c] * up[c];
            loss += fabsf(d) * mk[c] * norm;
        }
        const float k2 = r > 0.f ? dot / (r * n * n) : 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) g[c] = gc[c] / n - k2 * up[c];
    } else {
        const float lb = lab[0] >= 0.5f ? 1.f : 0.f;
        const float coef = (wneg * lb + (1.f - wneg) * (1.f - lb)) * norm;
        const float o = up[0], gz = o >= 0.f ? 1.f : 0.f;
        const float lv = o * (lb - gz) - log1pf(__expf(o - 2.f * o * gz));
        const float sg = 1.f / (1.f + __expf(-o));
        g[0] = -coef * (lb - sg);
        loss = -coef * lv;
    }
    return loss;
}

// weight of output index o for low-res index q along one axis (PyTorch's source index arithmetic)
__device__ __forceinline__ float up_weight(int o, int q, float rs, int n_in) {
    float s = ((float)o + 0.5f) * rs - 0.5f;
    s = s < 0.f ? 0.f : s;
    const int i0 = (int)s, i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    const float f = s - (float)i0;
    return (i0 == q ? 1.f - f : 0.f) + (i1 == q ? f : 0.f);
}

struct UpTile {
    int TQ, tiles_x, tiles_y;
};
static __host__ __device__ inline UpTile up_tile(int h, int w, int S, int TR) {
    UpTile t;
    t.TQ = 64 / S - 1;
    t.tiles_x = (w + t.TQ - 1) / t.TQ;
    t.tiles_y = (h + TR - 1) / TR;
    return t;
}

template <typename T, int KIND, int CMAX, int SC>
__global__ __launch_bounds__(64) void k_up_loss(const UpLossParams p) {
    extern __shared__ float sm[];
    constexpr int NL = KIND == 1 ? CMAX : 1;  // label values per pixel
    const int lane = threadIdx.x;
    const int S = SC ? SC : p.S;  // (compile-time for the scale the models run at)
    constexpr int TR = UL_TR;
    const UpTile tl = up_tile(p.h, p.w, S, TR);
    const int TQ = tl.TQ;
    const int b = blockIdx.x / (tl.tiles_x * tl.tiles_y);
    const int trem = blockIdx.x % (tl.tiles_x * tl.tiles_y);
    const int qy0 = (trem / tl.tiles_x) * TR, qx0 = (trem % tl.tiles_x) * TQ;
    const int C = p.C, Cs = C | 1;  // odd pixel stride
    const int RP = (TQ + 2) * Cs;
    const int H = p.h * S, W = p.w * S, S2 = 2 * S, hs = (S + 1) / 2;
    float* lowt = sm;                      // [TR + 2][TQ + 2][Cs]
    float* timg = lowt + (TR + 2) * RP;    // [64][Cs]   a completed low-res row, per output column
    float* wxt = timg + 64 * Cs;           // [TQ][2S]
    const T* low = reinterpret_cast<const T*>(p.low);
    const float rs = (float)p.h / (float)H;  // in / out, exactly 1/S

    for (int i = lane; i < (TR + 2) * (TQ + 2) * C; i += 64) {
        const int pix = i / C, c = i - pix * C;
        const int py = pix / (TQ + 2), px = pix - py * (TQ + 2);
        int gy = qy0 - 1 + py, gx = qx0 - 1 + px;
        gy = gy < 0 ? 0 : (gy > p.h - 1 ? p.h - 1 : gy);
        gx = gx < 0 ? 0 : (gx > p.w - 1 ? p.w - 1 : gx);
        lowt[py * RP + px * Cs + c] = mtl_to_f32(low[(((int64_t)b * p.h + gy) * p.w + gx) * C + c]);
    }
    // gather weights along x: entry (q, k) is the weight of output column S q - ceil(S/2) + k for low-res column q
    for (int i = lane; i < TQ * S2; i += 64) {
        const int ql = i / S2, k = i - ql * S2;
        const int ox = S * (qx0 + ql) - hs + k;
        wxt[i] = (ox >= 0 && ox < W && qx0 + ql < p.w) ? up_weight(ox, qx0 + ql, rs, p.w) : 0.f;
    }
    float norm;  // 1 / normaliser of the mean
    if (KIND == 0)
        norm = 1.f / p.stat[0];
    else if (KIND == 1)
        norm = 1.f / fmaxf(p.stat[0], 1e-6f);
    else
        norm = 1.f / ((float)p.B * (float)H * (float)W);
    const float wneg = KIND == 2 ? p.stat[0] : 0.f;

    // this lane's output column
    const int ox = S * qx0 - hs + lane;
    const bool col_ok = lane < (TQ + 1) * S && ox >= 0 && ox < W;
    const int oxc = ox < 0 ? 0 : (ox > W - 1 ? W - 1 : ox);
    float sx = ((float)oxc + 0.5f) * rs - 0.5f;
    sx = sx < 0.f ? 0.f : sx;
    const int ix0 = (int)sx, ix1 = ix0 + (ix0 < p.w - 1 ? 1 : 0);
    const float fx = sx - (float)ix0;
    // tile-local taps (clamped: a source index that rounds just outside the staged neighbourhood carries ~0 weight)
    int lx0 = ix0 - qx0 + 1, lx1 = ix1 - qx0 + 1;
    lx0 = lx0 < 0 ? 0 : (lx0 > TQ + 1 ? TQ + 1 : lx0);
    lx1 = lx1 < 0 ? 0 : (lx1 > TQ + 1 ? TQ + 1 : lx1);
    const int oqx = oxc / S - qx0;
    const bool own_x = col_ok && oqx >= 0 && oqx < TQ;
    const int64_t HW = (int64_t)H * W;
    const float* lbase = p.label + (int64_t)b * (KIND == 1 ? C : 1) * HW + oxc;

    const int oy_lo = S * qy0 - hs < 0 ? 0 : S * qy0 - hs;
    const int oy_hi = S * (qy0 + TR) + S / 2 > H ? H : S * (qy0 + TR) + S / 2;
    float a0[CMAX], a1[CMAX], v0[CMAX], v1[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) a0[c] = a1[c] = v0[c] = v1[c] = 0.f;
    float loss = 0.f;
    int cell = -1, cy1 = -1;  // source rows (i0, i1) of the current cell
    T* dl = reinterpret_cast<T*>(p.dlow);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // a completed low-res row q (its per-column sums in a0): gather along x, store
    auto emit = [&](int q) __attribute__((always_inline)) {
        if (q >= qy0 && q < qy0 + TR && q < p.h) {  // (wave-uniform)
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) timg[lane * Cs + c] = a0[c];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            for (int it = lane; it < TQ * C; it += 64) {
                const int ql = it / C, c = it - ql * C;
                const float* wq = wxt + ql * S2;
                const float* tc = timg + (S * ql) * Cs + c;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < S2; ++k) s += wq[k] * tc[k * Cs];
                if (qx0 + ql < p.w) dl[(((int64_t)b * p.h + q) * p.w + qx0 + ql) * C + c] = mtl_from_f32<T>(s);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // labels: PF rows are in flight while the previous PF rows are evaluated (one row ahead left the 1-channel head waiting on
    // every load: 3.5 waves per SIMD, ~0.1 us of arithmetic per row)
    constexpr int PF = CMAX > 8 ? 1 : 4;  // (the wide heads are arithmetic-bound and short of registers)
    float lab_next[PF][NL];
    auto fetch = [&](int oy0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int c = 0; c < NL; ++c)
                lab_next[u][c] = (col_ok && oy0 + u < oy_hi && c < C) ? lbase[(int64_t)c * HW + (int64_t)(oy0 + u) * W] : 0.f;
    };
    fetch(oy_lo);
    for (int oyb = oy_lo; oyb < oy_hi; oyb += PF) {
        float labs[PF][NL];
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int c = 0; c < NL; ++c) labs[u][c] = lab_next[u][c];
        if (oyb + PF < oy_hi) fetch(oyb + PF);
#pragma unroll 1  // (unrolled, the four rows' arithmetic interleaves: 256 VGPRs for the 21-class head)
        for (int u = 0; u < PF; ++u) {
        const int oy = oyb + u;
        if (oy >= oy_hi) break;
        float lab[NL];
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            lab[c] = labs[0][c];
#pragma unroll
            for (int v = 1; v < PF; ++v) lab[c] = u == v ? labs[v][c] : lab[c];  // (u is wave-uniform)
        }
        float sy = ((float)oy + 0.5f) * rs - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int iy0 = (int)sy, iy1 = iy0 + (iy0 < p.h - 1 ? 1 : 0);
        const float fy = sy - (float)iy0;
        if (iy0 != cell) {  // (wave-uniform) next source cell: row `cell` is complete, row cell + 1 carries over
            if (cell >= 0) {
                emit(cell);
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    a0[c] = (cy1 != cell && iy0 == cy1) ? a1[c] : 0.f;
                    a1[c] = 0.f;
                }
            }
            cell = iy0;
            cy1 = iy1;
            int ly0 = iy0 - qy0 + 1, ly1 = iy1 - qy0 + 1;
            ly0 = ly0 < 0 ? 0 : (ly0 > TR + 1 ? TR + 1 : ly0);
            ly1 = ly1 < 0 ? 0 : (ly1 > TR + 1 ? TR + 1 : ly1);
            const float* r00 = lowt + ly0 * RP + lx0 * Cs;
            const float* r01 = lowt + ly0 * RP + lx1 * Cs;
            const float* r10 = lowt + ly1 * RP + lx0 * Cs;
            const float* r11 = lowt + ly1 * RP + lx1 * Cs;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                v0[c] = c < C ? (1.f - fx) * r00[c] + fx * r01[c] : 0.f;
                v1[c] = c < C ? (1.f - fx) * r10[c] + fx * r11[c] : 0.f;
            }
        }
        float up[CMAX], g[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) up[c] = (1.f - fy) * v0[c] + fy * v1[c];
        const float lv = up_pixel<KIND, CMAX>(up, g, lab, C, p.ignore, norm, wneg);
        const int oqy = oy / S - qy0;  // the owner: the loss VALUE counts once over the tiles
        if (own_x && oqy >= 0 && oqy < TR) loss += lv;
        const float w0 = col_ok ? (iy1 == iy0 ? 1.f : 1.f - fy) : 0.f, w1 = (col_ok && iy1 != iy0) ? fy : 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            a0[c] += w0 * g[c];
            a1[c] += w1 * g[c];
        }
        }
    }
    if (cell >= 0) emit(cell);

    // loss value of the tile
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
    if (lane == 0) p.part[blockIdx.x] = loss;
}

template <typename T, int KIND, int CMAX, int SC>
static void launch_up_s(const UpLossParams& p, int64_t blocks, hipStream_t s) {
    const int cs = p.C | 1, tq = 64 / p.S - 1;
    const size_t lds = ((size_t)(UL_TR + 2) * (tq + 2) * cs + 64 * cs + (size_t)tq * 2 * p.S) * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;  // (one attribute call per instantiation)
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_loss<T, KIND, CMAX, SC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            raised = true;
        }
    }
    hipLaunchKernelGGL((k_up_loss<T, KIND, CMAX, SC>), dim3((unsigned)blocks), dim3(64), lds, s, p);
}
template <typename T, int KIND, int CMAX>
static void launch_up(const UpLossParams& p, int64_t blocks, hipStream_t s) {
    if (p.S == 8)  // the scale of the models' heads (56 -> 448, 28 -> 224)
        launch_up_s<T, KIND, CMAX, 8>(p, blocks, s);
    else
        launch_up_s<T, KIND, CMAX, 0>(p, blocks, s);
}

template <typename T>
static int dispatch_up(int kind, const UpLossParams& p, int64_t blocks, hipStream_t s) {
    if (kind == 0) {
        if (p.C <= 8)
            launch_up<T, 0, 8>(p, blocks, s);
        else if (p.C <= 24)
            launch_up<T, 0, 24>(p, blocks, s);
        else if (p.C <= 48)
            launch_up<T, 0, 48>(p, blocks, s);
        else
            return MTLORA_ERR_UNSUPPORTED;
    } else if (kind == 1) {
        if (p.C > 4) return MTLORA_ERR_UNSUPPORTED;
        launch_up<T, 1, 4>(p, blocks, s);
    } else if (kind == 2) {
        if (p.C != 1) return MTLORA_ERR_UNSUPPORTED;
        launch_up<T, 2, 1>(p, blocks, s);
    } else {
        return MTLORA_ERR_UNSUPPORTED;
    }
    return MTLORA_OK;
}

}  // namespace

extern "C" {

int64_t mtlora_upsample_loss_partials(int64_t B, int h, int w, int scale) {
    if (B < 0 || h <= 0 || w <= 0 || scale <= 0 || scale > 32) return -1;
    const UpTile t = up_tile(h, w, scale, UL_TR);
    return B * (int64_t)t.tiles_y * t.tiles_x;
}

int mtlora_upsample_loss(int kind, const void* low, const float* label, const float* stat, void* dlow, float* partials,
                         int64_t B, int h, int w, int C, int scale, int dtype, float ignore_index, void* stream) {
    if (B < 0 || h <= 0 || w <= 0 || C <= 0 || scale <= 0) return MTLORA_ERR_SHAPE;
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16) return MTLORA_ERR_DTYPE;
    if (B == 0) return MTLORA_OK;
    if (!low || !label || !stat || !dlow || !partials) return MTLORA_ERR_NULL;
    if (scale > 32) return MTLORA_ERR_UNSUPPORTED;
    const int64_t blocks = mtlora_upsample_loss_partials(B, h, w, scale);
    if (blocks >= ((int64_t)1 << 31) || (int64_t)h * scale * (int64_t)w * scale >= ((int64_t)1 << 31)) return MTLORA_ERR_SHAPE;
    UpLossParams p;
    p.low = low;
    p.label = label;
    p.stat = stat;
    p.dlow = dlow;
    p.part = partials;
    p.B = (int)B;
    p.h = h;
    p.w = w;
    p.C = C;
    p.S = scale;
    p.ignore = ignore_index;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MtlProfScope prof(PK_LOSS, (double)B * h * w * C * 2.0 * mtl_elem_size(dtype) +
                                   (double)B * h * scale * w * scale * 4.0 * (kind == 1 ? C : 1), s);
    const int rc = dtype == MTLORA_F32 ? dispatch_up<float>(kind, p, blocks, s) : dispatch_up<bf16>(kind, p, blocks, s);
    if (rc != MTLORA_OK) return rc;
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

}  // extern "C"
