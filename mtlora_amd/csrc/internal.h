// internal.h -- entry points shared between translation units of the library that are NOT part of the C ABI (hidden visibility:
// they do not appear among the .so's exports, include/mtlora_hip.h stays the whole boundary).
//
// The backward kernels of the LayerNorm family end in a small second-stage reduce launch (dgamma / dbeta from per-workgroup
// partials).  Nothing in the backward chain reads those results -- only the optimizer does -- so the one-call Swin block (block.hip)
// issues the main kernels in its phase 1 and the reduces together with the factor gradients in phase 2, on the side stream: `phase` 0 = kernel + reduce (what the public entry points do), 1 = main kernel only (the partials stay in `scratch`),
// 2 = the reduce of the partials only (same arguments).
#pragma once
#include <stdint.h>

#include "../../include/mtlora_hip.h"

#define MTL_INTERNAL extern "C" __attribute__((visibility("hidden")))

MTL_INTERNAL int mtli_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                    float* dgamma, float* dbeta, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                                    int64_t scratch_bytes, const void* dx_addend, int phase, void* stream);
MTL_INTERNAL int mtli_residual_layernorm_bwd(const void* dy, const void* x_new, const float* gamma, const float* mean, const float* rstd,
                                             void* d_shortcut, void* d_branch, float* dgamma, float* dbeta, const float* scale,
                                             int64_t B, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                                             int64_t scratch_bytes, const void* dx_addend, int phase, void* stream);

// k_hid_* (hid.hip): the task streams of a task-enabled Mlp without their hidden-width tensors; linear.hip fills the parameter blocks
struct HidLaunch;
struct HidParams;
struct HidRedParams;
MTL_INTERNAL void mtli_hid_launch(const HidLaunch* L, const HidParams* q, void* stream);
MTL_INTERNAL void mtli_hid_rows_finish(int dtype, const float* rowpart, int n_chunk, int64_t M, int nt, const float* alpha, const int* off,
                                       void* out, int ldo, void* stream);
MTL_INTERNAL void mtli_hid_reduce(const HidRedParams* r, int64_t per, void* stream);
