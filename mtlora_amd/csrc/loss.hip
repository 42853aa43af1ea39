// loss.hip -- the loss end of the train step (SURVEY 8 a13 callers: models/swin_mtl.py:245 final F.interpolate +
// mtl_loss_schemes.py), fused:   loss_t( interpolate(low, scale S, bilinear, align_corners=False), label_t )
// and d loss_t / d low in ONE kernel per task.  The (B, C, S*h, S*w) upsampled logits, their fp32 copy, the
// log-softmax / probabilities and the three gradient tensors of the same size that the ATen sequence writes and
// re-reads (7 passes over 540 MB for the 21-class head at B = 32, 448 x 448) never exist: the kernel reads the
// low-resolution logits (34 MB) and the labels (26 MB) and writes the low-resolution gradient.
//
// UNS threads per LOW-resolution pixel q (8 x 8 pixels per workgroup, their 10 x 10 neighbourhood staged in LDS as
// fp32; the threads of a pixel are adjacent lanes, take the output rows round-robin and add their shares with a
// fixed shuffle tree -- one thread per pixel left 1.5 waves per SIMD for 2S x 2S = 256 serial output pixels each at
// S = 8): a pixel visits the <= 2S x 2S output pixels whose bilinear taps include q, rebuilds their C interpolated logits
// (vertical lerp of the 3 neighbour columns once per output row, then a horizontal lerp), evaluates the per-pixel
// loss gradient and accumulates  w_y w_x d loss / d up  -- the transpose of the interpolation as a GATHER, so the
// result is deterministic (no float atomics).  Every output pixel is evaluated by the <= 4 low-res pixels it taps
// (4x redundant transcendental work, ~0.1 ms per task) and contributes to the loss VALUE only from its owner
// q = (oy / S, ox / S).  Index / weight arithmetic is PyTorch's (area_pixel_compute_source_index, align_corners=False:
// src = max((dst + 0.5) * in/out - 0.5, 0), i1 = i0 + (i0 < in - 1)), evaluated per output pixel, so the clamped
// borders need no special case.
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

constexpr int UQ = 8;   // low-res pixels per workgroup side
constexpr int UNS = 4;  // threads per low-res pixel (power of two, <= 64)

template <typename T, int KIND, int CMAX>
__global__ __launch_bounds__(UQ * UQ * UNS) void k_up_loss(const UpLossParams p) {
    extern __shared__ float sm[];  // [(UQ+2)*(UQ+2)][Cs]
    constexpr int NT = UQ * UQ * UNS;
    __shared__ float red[NT / 64];
    const int tid = threadIdx.x;
    const int tiles_x = (p.w + UQ - 1) / UQ, tiles_y = (p.h + UQ - 1) / UQ;
    const int b = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x % (tiles_x * tiles_y);
    const int qy0 = (trem / tiles_x) * UQ, qx0 = (trem % tiles_x) * UQ;
    const int C = p.C, Cs = C | 1;  // odd pixel stride: the 16 pixels of a tile row hit 16 different banks
    // tile-row pitch == 16 (mod 32) words: the two tile rows a 32-lane half touches land in the complementary banks
    const int RP = (UQ + 2) * Cs + ((16 - (UQ + 2) * Cs) & 31);
    const int H = p.h * p.S, W = p.w * p.S, S = p.S;
    const T* low = reinterpret_cast<const T*>(p.low);

    for (int i = tid; i < (UQ + 2) * (UQ + 2) * C; i += NT) {
        const int pix = i / C, c = i - pix * C;
        int gy = qy0 - 1 + pix / (UQ + 2), gx = qx0 - 1 + pix % (UQ + 2);
        gy = gy < 0 ? 0 : (gy > p.h - 1 ? p.h - 1 : gy);
        gx = gx < 0 ? 0 : (gx > p.w - 1 ? p.w - 1 : gx);
        sm[(pix / (UQ + 2)) * RP + (pix % (UQ + 2)) * Cs + c] = mtl_to_f32(low[(((int64_t)b * p.h + gy) * p.w + gx) * C + c]);
    }
    __syncthreads();

    const int px = tid / UNS, part = tid % UNS;
    const int ly = px / UQ, lx = px % UQ;
    const int qy = qy0 + ly, qx = qx0 + lx;
    const bool active = qy < p.h && qx < p.w;
    float g[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) g[c] = 0.f;
    float loss = 0.f;

    if (active) {
        const float rs = (float)p.h / (float)H;  // in / out, exactly 1/S
        float norm;                              // 1 / normaliser of the mean
        if (KIND == 0)
            norm = 1.f / p.stat[0];
        else if (KIND == 1)
            norm = 1.f / fmaxf(p.stat[0], 1e-6f);
        else
            norm = 1.f / ((float)p.B * (float)H * (float)W);
        const float wneg = KIND == 2 ? p.stat[0] : 0.f;
        const int oy_lo = S * qy - S < 0 ? 0 : S * qy - S, oy_hi = S * qy + 2 * S > H ? H : S * qy + 2 * S;
        const int ox_lo = S * qx - S < 0 ? 0 : S * qx - S, ox_hi = S * qx + 2 * S > W ? W : S * qx + 2 * S;
        for (int oy = oy_lo + part; oy < oy_hi; oy += UNS) {
            float sy = ((float)oy + 0.5f) * rs - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            const int iy0 = (int)sy, iy1 = iy0 + (iy0 < p.h - 1 ? 1 : 0);
            const float fy = sy - (float)iy0;
            const float wy = (iy0 == qy ? 1.f - fy : 0.f) + (iy1 == qy ? fy : 0.f);
            if (wy == 0.f) continue;
            const float* r0 = sm + (iy0 - qy0 + 1) * RP + lx * Cs;
            const float* r1 = sm + (iy1 - qy0 + 1) * RP + lx * Cs;
            float v[3][CMAX];  // vertical lerp at columns qx-1, qx, qx+1
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < CMAX; ++c) v[j][c] = c < C ? (1.f - fy) * r0[j * Cs + c] + fy * r1[j * Cs + c] : 0.f;
            for (int ox = ox_lo; ox < ox_hi; ++ox) {
                float sx = ((float)ox + 0.5f) * rs - 0.5f;
                sx = sx < 0.f ? 0.f : sx;
                const int ix0 = (int)sx, ix1 = ix0 + (ix0 < p.w - 1 ? 1 : 0);
                const float fx = sx - (float)ix0;
                const float wx = (ix0 == qx ? 1.f - fx : 0.f) + (ix1 == qx ? fx : 0.f);
                if (wx == 0.f) continue;
                // horizontal lerp as a 3-weight blend of the neighbour columns (no per-channel selects)
                const int j0 = ix0 - qx + 1, j1 = ix1 - qx + 1;
                const float b0 = (j0 == 0 ? 1.f - fx : 0.f) + (j1 == 0 ? fx : 0.f);
                const float b1 = (j0 == 1 ? 1.f - fx : 0.f) + (j1 == 1 ? fx : 0.f);
                const float b2 = (j0 == 2 ? 1.f - fx : 0.f) + (j1 == 2 ? fx : 0.f);
                float up[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) up[c] = b0 * v[0][c] + b1 * v[1][c] + b2 * v[2][c];
                const float wq = wy * wx;
                const bool own = (oy / S == qy) && (ox / S == qx);
                const int64_t lpix = (int64_t)oy * W + ox;
                if (KIND == 0) {
                    const float lab = p.label[(int64_t)b * H * W + lpix];
                    if (lab != p.ignore) {
                        const int cls = (int)lab;
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
                            g[c] += wq * norm * (pc - (c == cls ? 1.f : 0.f));
                        }
                        if (own) loss -= __logf(ucls) * norm;
                    }
                } else if (KIND == 1) {
                    float l[CMAX], mk[CMAX];
                    float r2 = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        l[c] = c < C ? p.label[((int64_t)b * C + c) * H * W + lpix] : 0.f;
                        mk[c] = (c < C && l[c] != p.ignore) ? 1.f : 0.f;
                        r2 += up[c] * up[c];
                    }
                    const float r = sqrtf(r2), n = r + 1e-12f;
                    float gc[CMAX], dot = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        const float d = up[c] / n - l[c];
                        gc[c] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * mk[c] * norm;
                        dot += gc[c] * up[c];
                        if (own) loss += fabsf(d) * mk[c] * norm;
                    }
                    const float k2 = r > 0.f ? dot / (r * n * n) : 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) g[c] += wq * (gc[c] / n - k2 * up[c]);
                } else {
                    const float lab = p.label[(int64_t)b * H * W + lpix] >= 0.5f ? 1.f : 0.f;
                    const float coef = (wneg * lab + (1.f - wneg) * (1.f - lab)) * norm;
                    const float o = up[0], gz = o >= 0.f ? 1.f : 0.f;
                    const float lv = o * (lab - gz) - log1pf(__expf(o - 2.f * o * gz));
                    const float sg = 1.f / (1.f + __expf(-o));
                    g[0] += wq * (-coef * (lab - sg));
                    if (own) loss -= coef * lv;
                }
            }
        }
    }
    // the UNS shares of a pixel (adjacent lanes; inactive pixels hold zeros) -> fixed butterfly, then lane `part == 0` writes
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
        for (int o = 1; o < UNS; o <<= 1) g[c] += __shfl_xor(g[c], o);
    if (active && part == 0) {
        T* dl = reinterpret_cast<T*>(p.dlow) + (((int64_t)b * p.h + qy) * p.w + qx) * C;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) dl[c] = mtl_from_f32<T>(g[c]);
    }

    // loss value: wave reduce, then the waves in a fixed order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
    if ((tid & 63) == 0) red[tid >> 6] = loss;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        p.part[blockIdx.x] = t;
    }
}

template <typename T, int KIND, int CMAX>
static void launch_up(const UpLossParams& p, int64_t blocks, hipStream_t s) {
    const int cs = p.C | 1, rp = (UQ + 2) * cs + ((16 - (UQ + 2) * cs) & 31);
    const size_t lds = (size_t)(UQ + 2) * rp * sizeof(float);
    hipLaunchKernelGGL((k_up_loss<T, KIND, CMAX>), dim3((unsigned)blocks), dim3(UQ * UQ * UNS), lds, s, p);
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

int64_t mtlora_upsample_loss_partials(int64_t B, int h, int w) {
    if (B < 0 || h <= 0 || w <= 0) return -1;
    return B * (int64_t)((h + UQ - 1) / UQ) * ((w + UQ - 1) / UQ);
}

int mtlora_upsample_loss(int kind, const void* low, const float* label, const float* stat, void* dlow, float* partials,
                         int64_t B, int h, int w, int C, int scale, int dtype, float ignore_index, void* stream) {
    if (B < 0 || h <= 0 || w <= 0 || C <= 0 || scale <= 0) return MTLORA_ERR_SHAPE;
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16) return MTLORA_ERR_DTYPE;
    if (B == 0) return MTLORA_OK;
    if (!low || !label || !stat || !dlow || !partials) return MTLORA_ERR_NULL;
    const int64_t blocks = mtlora_upsample_loss_partials(B, h, w);
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
