// window_process.hip -- cyclic shift + window partition / window merge + reverse shift as single-pass
// gather copies (CDNA4).  Replaces kernels/window_process/swin_window_process_kernel.cu:42-147.
//
// Design (HBM-bound, zero arithmetic): one thread moves one 16/8/4/2-byte unit; a token row of C
// elements is contiguous in both tensors, so consecutive threads read and write consecutive
// addresses (coalesced 1 KiB per wave instruction at 16 B/lane).  The output index is linear in
// the thread id (perfectly coalesced stores); the input index is the gathered one.
//
// Two primitive maps cover the four reference entry points:
//   image -> windows:  win[(b,wy,wx), ty, tx, :] = img[b, (wy*ws+ty + d) mod H, (wx*ws+tx + d) mod W, :]
//   windows -> image:  img[b, y, x, :] = win[(b, ys/ws, xs/ws), ys%ws, xs%ws, :],  ys=(y+d) mod H, xs=(x+d) mod W
#include "common.h"

namespace {

template <typename U>
__global__ __launch_bounds__(256) void k_image_to_windows(const U* __restrict__ img, U* __restrict__ win, int64_t total,
                                                          int H, int W, int cu /* units per token */, int ws, int d) {
    const int nWx = W / ws, nWy = H / ws;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(o % cu);
        int64_t t = o / cu;  // output token index in window order
        int tx = (int)(t % ws);
        t /= ws;
        int ty = (int)(t % ws);
        t /= ws;
        int wx = (int)(t % nWx);
        t /= nWx;
        int wy = (int)(t % nWy);
        int64_t b = t / nWy;
        int y = ((wy * ws + ty + d) % H + H) % H;
        int x = ((wx * ws + tx + d) % W + W) % W;
        win[o] = img[((b * H + y) * W + x) * cu + c];
    }
}

template <typename U>
__global__ __launch_bounds__(256) void k_windows_to_image(const U* __restrict__ win, U* __restrict__ img, int64_t total,
                                                          int H, int W, int cu, int ws, int d) {
    const int nWx = W / ws, nWy = H / ws;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(o % cu);
        int64_t t = o / cu;  // output token index in image order
        int x = (int)(t % W);
        t /= W;
        int y = (int)(t % H);
        int64_t b = t / H;
        int ys = ((y + d) % H + H) % H;
        int xs = ((x + d) % W + W) % W;
        int64_t w = (b * nWy + ys / ws) * nWx + xs / ws;
        img[o] = win[((w * ws + ys % ws) * ws + xs % ws) * cu + c];
    }
}

struct U16 {
    uint32_t a, b, c, d;
};
struct U8 {
    uint32_t a, b;
};

int launch(bool to_windows, const void* src, void* dst, int64_t B, int64_t H, int64_t W, int64_t C, int d, int ws,
           int dtype, void* stream) {
    if (!src || !dst) return MTLORA_ERR_NULL;
    if (dtype != MTLORA_F32 && dtype != MTLORA_BF16 && dtype != MTLORA_F16) return MTLORA_ERR_DTYPE;
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || ws <= 0 || H % ws || W % ws) return MTLORA_ERR_SHAPE;
    if (B * H * W * C >= ((int64_t)1 << 40)) return MTLORA_ERR_SHAPE;
    if (B == 0) return MTLORA_OK;
    const int es = mtl_elem_size(dtype);
    const int64_t row_bytes = C * es;
    const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes;
    int unit = (al % 16 == 0) ? 16 : (al % 8 == 0) ? 8 : (al % 4 == 0) ? 4 : 2;
    if (es == 4 && unit < 4) return MTLORA_ERR_ALIGN;
    const int cu = (int)(row_bytes / unit);
    const int64_t total = B * H * W * cu;
    int64_t blocks = mtl_ceil_div(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride the rest (G11)
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)blocks), blk(256);
    MtlProfScope prof(PK_WINDOW, 2.0 * (double)total * unit, s);
#define MTL_WP(U)                                                                                                     \
    if (to_windows)                                                                                                   \
        hipLaunchKernelGGL(k_image_to_windows<U>, g, blk, 0, s, (const U*)src, (U*)dst, total, (int)H, (int)W, cu,    \
                           ws, d);                                                                                    \
    else                                                                                                              \
        hipLaunchKernelGGL(k_windows_to_image<U>, g, blk, 0, s, (const U*)src, (U*)dst, total, (int)H, (int)W, cu,    \
                           ws, d);
    switch (unit) {
        case 16: MTL_WP(U16) break;
        case 8: MTL_WP(U8) break;
        case 4: MTL_WP(uint32_t) break;
        default: MTL_WP(uint16_t) break;
    }
#undef MTL_WP
    MTL_CHECK_LAUNCH();
    return MTLORA_OK;
}

}  // namespace

extern "C" {

// swin_window_process.cpp:70 / .cu:42 : out[win] = in[(pos - shift_size) mod]
int mtlora_roll_and_window_partition_forward(const void* image, void* windows, int64_t B, int64_t H, int64_t W,
                                             int64_t C, int shift_size, int window_size, int dtype, void* stream) {
    return launch(true, image, windows, B, H, W, C, -shift_size, window_size, dtype, stream);
}
// .cu:69 : grad_out[img] = grad_in[win of (pos + shift_size) mod]
int mtlora_roll_and_window_partition_backward(const void* grad_windows, void* grad_image, int64_t B, int64_t H,
                                              int64_t W, int64_t C, int shift_size, int window_size, int dtype,
                                              void* stream) {
    return launch(false, grad_windows, grad_image, B, H, W, C, shift_size, window_size, dtype, stream);
}
// .cu:96 : out[img] = in[win of (pos - shift_size) mod]
int mtlora_window_merge_and_roll_forward(const void* windows, void* image, int64_t B, int64_t H, int64_t W,
                                         int64_t C, int shift_size, int window_size, int dtype, void* stream) {
    return launch(false, windows, image, B, H, W, C, -shift_size, window_size, dtype, stream);
}
// .cu:124 : grad_out[win] = grad_in[(pos + shift_size) mod]
int mtlora_window_merge_and_roll_backward(const void* grad_image, void* grad_windows, int64_t B, int64_t H,
                                          int64_t W, int64_t C, int shift_size, int window_size, int dtype,
                                          void* stream) {
    return launch(true, grad_image, grad_windows, B, H, W, C, shift_size, window_size, dtype, stream);
}

int mtlora_version(void) { return MTLORA_ABI_VERSION; }

const char* mtlora_error_string(int status) {
    switch (status) {
        case MTLORA_OK: return "ok";
        case MTLORA_ERR_DTYPE: return "unsupported dtype";
        case MTLORA_ERR_SHAPE: return "bad shape";
        case MTLORA_ERR_ALIGN: return "misaligned pointer or leading dimension";
        case MTLORA_ERR_NULL: return "null pointer";
        case MTLORA_ERR_WORKSPACE: return "ctx/scratch buffer too small";
        case MTLORA_ERR_HIP: return "HIP launch error";
        case MTLORA_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
}
