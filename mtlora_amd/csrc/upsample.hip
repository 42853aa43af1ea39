// upsample.hip -- channels-last bilinear upsampling by an integer factor (align_corners=False), forward and backward,
// for the HRNet head of the callers (models/seg_hrnet.py:498-526: F.interpolate of the three coarse maps to the finest
// resolution, then torch.cat along channels).  Both kernels address the FINE tensor through a row stride and a channel
// offset, so the forward writes each upsampled map directly into its channel slice of the concatenated (pixels, C_total)
// matrix the 1x1 convolution consumes, and the backward reads that slice of the matrix' gradient: the concat / slice
// copies disappear.  Backward is the transpose of the interpolation written as a GATHER over the <= (2S)^2 fine pixels
// that tap a coarse pixel (deterministic; ATen's nhwc backward takes 135 us per map at B = 32).
// Index / weight arithmetic is PyTorch's area_pixel_compute_source_index (see loss.hip).  4 channels per thread.
#include "common.h"

namespace {

struct UpParams {
    const void* src;
    void* dst;
    int B, h, w, C, S;   // coarse (B, h, w, C); fine (B, S*h, S*w, .)
    int64_t ld_fine;     // elements per fine pixel row (>= C), channel offset already applied to the pointer
};

template <typename T>
struct V4;
template <>
struct V4<float> {
    typedef f32x4 raw;
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <>
struct V4<bf16> {
    static __device__ __forceinline__ f32x4 ld(const bf16* p) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void st(bf16* p, f32x4 v) {
        *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
};

__device__ __forceinline__ void src_index(int o, float rs, int n_in, int& i0, int& i1, float& f) {
    float s = ((float)o + 0.5f) * rs - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    f = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_up_fwd(const UpParams p) {
    const int H = p.h * p.S, W = p.w * p.S, CV = p.C / 4;
    const float rs = 1.f / (float)p.S;
    const T* src = reinterpret_cast<const T*>(p.src);
    T* dst = reinterpret_cast<T*>(p.dst);
    const int64_t total = (int64_t)p.B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int ox = (int)(r % W);
        r /= W;
        const int oy = (int)(r % H);
        const int64_t b = r / H;
        int iy0, iy1, ix0, ix1;
        float fy, fx;
        src_index(oy, rs, p.h, iy0, iy1, fy);
        src_index(ox, rs, p.w, ix0, ix1, fx);
        const T* base = src + b * p.h * p.w * p.C + cv * 4;
        const f32x4 a = V4<T>::ld(base + ((int64_t)iy0 * p.w + ix0) * p.C), bq = V4<T>::ld(base + ((int64_t)iy0 * p.w + ix1) * p.C);
        const f32x4 c = V4<T>::ld(base + ((int64_t)iy1 * p.w + ix0) * p.C), d = V4<T>::ld(base + ((int64_t)iy1 * p.w + ix1) * p.C);
        const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = w00 * a[e] + w01 * bq[e] + w10 * c[e] + w11 * d[e];
        V4<T>::st(dst + ((b * H + oy) * W + ox) * p.ld_fine + cv * 4, o);
    }
}

// src = gradient of the fine tensor (strided), dst = gradient of the coarse tensor.  UB adjacent lanes share one (coarse
// pixel, 4-channel vector): they take the <= 2S tapping fine rows round-robin and add their shares with a fixed shuffle
// tree (one thread per output left a 64-iteration serial gather of 8-byte loads per thread at S = 4).
constexpr int UB = 4;
template <typename T>
__global__ __launch_bounds__(256) void k_up_bwd(const UpParams p) {
    const int H = p.h * p.S, W = p.w * p.S, CV = p.C / 4, S = p.S;
    const float rs = 1.f / (float)S;
    const T* g = reinterpret_cast<const T*>(p.src);
    T* dst = reinterpret_cast<T*>(p.dst);
    const int64_t total = (int64_t)p.B * p.h * p.w * CV;
    const int part = threadIdx.x % UB;
    // (all lanes of a shuffle group run the same number of iterations: the loop bound is rounded up per group)
    const int64_t stride = (int64_t)gridDim.x * (256 / UB);
    for (int64_t i0 = (int64_t)blockIdx.x * (256 / UB); i0 < total; i0 += stride) {
        const int64_t i = i0 + threadIdx.x / UB;
        const bool live = i < total;
        const int64_t ii = live ? i : 0;
        const int cv = (int)(ii % CV);
        int64_t r = ii / CV;
        const int qx = (int)(r % p.w);
        r /= p.w;
        const int qy = (int)(r % p.h);
        const int64_t b = r / p.h;
        // the fine pixels that tap coarse index q along an axis are S q - ceil(S / 2) ... + 2S - 1 (loss.hip); one more on either side
        // covers a source index that rounds across a cell boundary (its weight is then ~0 and the exact test below drops or keeps it)
        const int hs1 = (S + 1) / 2 + 1;
        const int oy_lo = S * qy - hs1 < 0 ? 0 : S * qy - hs1, oy_hi = S * qy - hs1 + 2 * S + 2 > H ? H : S * qy - hs1 + 2 * S + 2;
        const int ox_lo = S * qx - hs1 < 0 ? 0 : S * qx - hs1, ox_hi = S * qx - hs1 + 2 * S + 2 > W ? W : S * qx - hs1 + 2 * S + 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            for (int oy = oy_lo + part; oy < oy_hi; oy += UB) {
                int iy0, iy1;
                float fy;
                src_index(oy, rs, p.h, iy0, iy1, fy);
                const float wy = (iy0 == qy ? 1.f - fy : 0.f) + (iy1 == qy ? fy : 0.f);
                if (wy == 0.f) continue;
                const T* row = g + ((b * H + oy) * W) * p.ld_fine + cv * 4;
                for (int ox = ox_lo; ox < ox_hi; ++ox) {
                    int ix0, ix1;
                    float fx;
                    src_index(ox, rs, p.w, ix0, ix1, fx);
                    const float wq = wy * ((ix0 == qx ? 1.f - fx : 0.f) + (ix1 == qx ? fx : 0.f));
                    if (wq == 0.f) continue;
                    const f32x4 v = V4<T>::ld(row + (int64_t)ox * p.ld_fine);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += wq * v[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = 1; o < UB; o <<= 1) acc[e] += __shfl_xor(acc[e], o);
        if (live && part == 0) V4<T>::st(dst + ((b * p.h + qy) * p.w + qx) * p.C + cv * 4, acc);
    }
}

static int check_up(const void* a, const void* b, int64_t B, int h, int w, int C, int S, int64_t ld, int dtype) {
    if (B < 0 || h <= 0 || w <= 0 || C <= 0 || S <= 0 || ld < C) return MTLORA_ERR_SHAPE;
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16) return MTLORA_ERR_DTYPE;
    if (C % 4 || ld % 4) return MTLORA_ERR_ALIGN;
    if (B > 0 && (!a || !b)) return MTLORA_ERR_NULL;
    const size_t al = dtype == MTLORA_F32 ? 16 : 8;
    if (((uintptr_t)a % al) || ((uintptr_t)b % al)) return MTLORA_ERR_ALIGN;
    if (B * (int64_t)h * S * w * S >= ((int64_t)1 << 40)) return MTLORA_ERR_SHAPE;
    return MTLORA_OK;
}

}  // namespace

extern "C" {

int mtlora_upsample_cl_fwd(const void* coarse, void* fine, int64_t B, int h, int w, int C, int scale, int64_t ld_fine,
                           int dtype, void* stream) {
    const int rc = check_up(coarse, fine, B, h, w, C, scale, ld_fine, dtype);
    if (rc != MTLORA_OK || B == 0) return rc;
    UpParams p = {coarse, fine, (int)B, h, w, C, scale, ld_fine};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = B * (int64_t)h * scale * w * scale * (C / 4);
    int64_t blocks = mtl_ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    MtlProfScope prof(PK_UPSAMPLE, (double)mtl_elem_size(dtype) * B * h * w * C * (1.0 + (double)scale * scale), s);
    if (dtype == MTLORA_F32)
        hipLaunchKernelGGL(k_up_fwd<float>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(k_up_fwd<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

int mtlora_upsample_cl_bwd(const void* grad_fine, void* grad_coarse, int64_t B, int h, int w, int C, int scale,
                           int64_t ld_fine, int dtype, void* stream) {
    const int rc = check_up(grad_fine, grad_coarse, B, h, w, C, scale, ld_fine, dtype);
    if (rc != MTLORA_OK || B == 0) return rc;
    UpParams p = {grad_fine, grad_coarse, (int)B, h, w, C, scale, ld_fine};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = B * (int64_t)h * w * (C / 4);
    int64_t blocks = mtl_ceil_div(total, 256 / UB);
    if (blocks > 65536) blocks = 65536;
    MtlProfScope prof(PK_UPSAMPLE, (double)mtl_elem_size(dtype) * B * h * w * C * (1.0 + (double)scale * scale), s);
    if (dtype == MTLORA_F32)
        hipLaunchKernelGGL(k_up_bwd<float>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(k_up_bwd<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

}  // extern "C"
