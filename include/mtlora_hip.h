/*
 * mtlora_hip.h -- C ABI of libmtlora_hip.so, the MI355X (gfx950) implementation of the
 * MTLoRA data-parallel hot path.
 *
 * Boundary contract:
 *   - plain C, raw DEVICE pointers + sizes + a dtype enum + a hipStream_t passed as void*;
 *     no torch / ATen / pybind types anywhere;
 *   - the CALLER owns every buffer (inputs, outputs, ctx, scratch); the library never
 *     allocates, frees, retains a pointer or synchronises; every launch goes on `stream`;
 *   - re-entrant (safe from the autograd thread and across DDP ranks): nothing a call computes depends on state left by
 *     another call.  The only process-wide state is a per-device cache of device facts (CU count, "dynamic LDS limit raised"
 *     marks) and the opt-in profiler of mtlora_prof_begin/end; kernel selection is a function of the descriptor alone
 *     (ABI v6: no environment variable influences what the library computes or which kernel it picks; the
 *     opt-in profiler alone reads MTLORA_PROF_DUMP, a file name for its record dump, in mtlora_prof_end);
 *   - every function returns MTLORA_OK (0) or a negative mtlora_status; the Python shim
 *     (mtlora_amd/_lib.py) turns a non-zero status into RuntimeError -- the same observable
 *     behaviour as the reference's AT_ASSERTM -> RuntimeError (swin_window_process.cpp:64-66).
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef MTLORA_HIP_H
#define MTLORA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTLORA_ABI_VERSION 8
#define MTLORA_MAX_TASKS 8

typedef enum mtlora_dtype {
    MTLORA_F32 = 0,  /* exact-f32 MFMA path (v_mfma_f32_32x32x2_f32) */
    MTLORA_BF16 = 1, /* bf16 in/out, fp32 accumulate (v_mfma_f32_32x32x16_bf16) */
    MTLORA_F16 = 2   /* fp16 in/out, fp32 accumulate (v_mfma_f32_32x32x16_f16): MTLoRALinear, window attention, gemm_tn, the
                        window_process copies and the block glue (LayerNorm family, residual + DropPath, BatchNorm + ReLU) -- the
                        reference's default autocast dtype (main.py:341); upsample / loss / column-sum entries take fp32 / bf16 */
} mtlora_dtype;

typedef enum mtlora_status {
    MTLORA_OK = 0,
    MTLORA_ERR_DTYPE = -1,
    MTLORA_ERR_SHAPE = -2,
    MTLORA_ERR_ALIGN = -3,
    MTLORA_ERR_NULL = -4,
    MTLORA_ERR_WORKSPACE = -5,
    MTLORA_ERR_HIP = -6,
    MTLORA_ERR_UNSUPPORTED = -7
} mtlora_status;

/* ABI version of the loaded library (== MTLORA_ABI_VERSION of the header it was built from). */
int mtlora_version(void);
const char* mtlora_error_string(int status);

/* ------------------------------------------------------------------------------------------
 * Window process -- replaces the pybind module `swin_window_process`
 * (kernels/window_process/swin_window_process.cpp:127-132) and its four CUDA kernels
 * (swin_window_process_kernel.cu:42,69,96,124).  Same argument meaning and SIGN CONVENTION:
 *   WindowProcess.apply(x,B,H,W,C,-shift,ws)         -> *_partition_forward (shift_size = -shift)
 *   WindowProcessReverse.apply(w,B,H,W,C,+shift,ws)  -> *_merge_and_roll_forward (shift_size = +shift)
 * Differences: output is caller-allocated; bf16 supported (reference: fp16/fp32 only,
 * .cu:170-175); correct for H != W (reference .cu:113-115 is only exact for square maps);
 * launches on `stream` (reference: default stream, .cu:178).
 * `image` is (B,H,W,C) contiguous, `windows` is (B*(H/ws)*(W/ws), ws, ws, C) contiguous.
 * ------------------------------------------------------------------------------------------ */
int mtlora_roll_and_window_partition_forward(const void* image, void* windows, int64_t B, int64_t H, int64_t W,
                                             int64_t C, int shift_size, int window_size, int dtype, void* stream);
int mtlora_roll_and_window_partition_backward(const void* grad_windows, void* grad_image, int64_t B, int64_t H,
                                              int64_t W, int64_t C, int shift_size, int window_size, int dtype,
                                              void* stream);
int mtlora_window_merge_and_roll_forward(const void* windows, void* image, int64_t B, int64_t H, int64_t W,
                                         int64_t C, int shift_size, int window_size, int dtype, void* stream);
int mtlora_window_merge_and_roll_backward(const void* grad_image, void* grad_windows, int64_t B, int64_t H,
                                          int64_t W, int64_t C, int shift_size, int window_size, int dtype,
                                          void* stream);

/* ------------------------------------------------------------------------------------------
 * MTLoRALinear -- replaces the ATen op sequence of models/lora.py:253-284 (forward) and its
 * autograd backward:
 *   Y_s = X W^T + b + s_s (D(X) A_s^T) B_s^T
 *   Y_t = X W^T + b [+ s_s(...) if mode==matrixv2] + s_t (X_t A_t^T) B_t^T,   X_t = x_tasks[t] or D(X)
 * D = dropout(p) (train only; counter-based generator `mtl_dropout_keep`, restated in
 * oracle/mtlora_oracle.py:dropout_keep_mask).  'addition' mode = r_s 0 + T>0 here; its
 * LayerNorm(sum_t) tail is applied by the host module.
 *
 * Layouts: X, X_t, Y_*: (M,K)/(M,N) row-major in `dtype`.  W (N,K), bias (N): `dtype` / fp32.
 * LoRA masters A_* (r,K), B_* (N,r): fp32 (the nn.Parameters themselves).  Parameter
 * gradients are fp32.  Wt is W transposed, (K,N) row-major in `dtype` (W is frozen; the host
 * caches it).
 * ------------------------------------------------------------------------------------------ */
typedef struct mtlora_linear_desc {
    int64_t M, K, N;
    int32_t dtype;      /* MTLORA_F32 | MTLORA_BF16 | MTLORA_F16 */
    int32_t mode;       /* 0 = 'matrix', 1 = 'matrixv2' (lora.py:259-274) */
    int32_t T;          /* number of task outputs, 0..MTLORA_MAX_TASKS (0 <=> tasks is None) */
    int32_t r_s;        /* shared rank, 0 = no shared update */
    int32_t r_t[MTLORA_MAX_TASKS];
    float scale_s;
    float scale_t[MTLORA_MAX_TASKS];
    int32_t has_x_tasks; /* 1: task t reads x_t[t] undropped; 0: task t reads D(X) (lora.py:262-263) */
    float dropout_p;     /* 0 in eval */
    uint64_t seed;       /* dropout seed of this call (same value for fwd and bwd) */
    const uint64_t* seed_offset; /* optional DEVICE pointer (null = none): the kernels use seed + *seed_offset (mod 2^64),
                                    read when they run -- a captured HIP graph draws fresh masks on every replay by
                                    bumping that one device word between replays (ABI v2) */
    int32_t bwd_phase;   /* backward only (ABI v3).  0: everything on `stream`.  1: dX / dX_t only (and the Q = alpha dY B
                            scratch the factor gradients need); 2: the factor gradients dA_* / dB_* only -- lets the caller
                            put them on a second stream next to the rest of the backward chain: call phase 1 on stream s1,
                            order s2 after it (event), call phase 2 with the SAME arguments on s2, and join s2 before the
                            gradients are read.  Nothing but dA / dB depends on phase 2. */
    /* ---- kernel selection (ABI v6; all 0 = the library's own choice).  Results are the same whatever is selected (the
     * kernel families are bit-compatible up to fp32 summation order); tests pin every family against the oracle by forcing
     * it on shapes far below the sizes at which the library would pick it, and A/B timing uses the same switches. */
    int32_t sel_stream;  /* wave-streaming kernels (csrc/stream.h): 0 wherever eligible, 1 never (tiled kernels only) */
    int32_t sel_dense;   /* k_ntd / k_nte (csrc/dense.h), single-output MFMA-dense launches: 0 by shape heuristics, 1 never, 2 k_ntd whenever
                            eligible, 3 k_nte (two 4-wave workgroups per CU) whenever eligible, 4 heuristics without k_nte (A/B) */
    int32_t sel_tn;      /* k_sp_tn, streaming factor gradients: 0 when every wave gets >= 8 slabs, 1 never, 2 whenever eligible */
    int32_t sel_projk;   /* P / Q passes whose projection rows do not fit in LDS: 0 heuristics (k_sp_projk on single-round launches, k_pq on
                            single-source passes, tiled otherwise), 1 tiled only, 2 k_sp_projk whenever eligible, 3 k_pq whenever eligible */
    int32_t max_cu;      /* 0: size persistent grids for the whole device; n > 0: as if the device had n CUs -- every wave /
                            workgroup of a persistent kernel then owns MANY work items even at test sizes (the steady state
                            of the slot rings and the vmcnt accounting, reached otherwise only at benchmark sizes) */
    const void* packed;  /* ABI v6, optional DEVICE pointer (null = none): the layer's packed low-rank factors, written earlier by
                            mtlora_linear_pack / mtlora_linear_pack_table from the SAME masters, scales, dropout_p and has_x_tasks.
                            fwd then skips its own packing launch and ctx only holds P; bwd must get the same pointer.  The masters
                            change once per optimizer step, so a trainer packs every layer in ONE launch per step instead of one
                            launch per layer and call. */
    int32_t hid;         /* ABI v8, 0 = none: role of this call inside an Mlp whose TASK hidden tensors stay implicit (MTLORA_HID_*, see
                            mtlora_mlp_hid_* below) */
    int32_t hid_pad_;
    const void* hid_ptr; /* ABI v8: MTLORA_HID_FWD_BASE: OUT (M x N), the pretrained product x W^T + b without any low-rank update;
                            MTLORA_HID_Q_GIVEN: IN (M x N), the summed output gradient G the dense part of dX is formed from (nullable) */
} mtlora_linear_desc;

/* d->hid flags (ABI v8) */
#define MTLORA_HID_FWD_BASE 1 /* fwd (the Mlp's fc1): writes y_s (+ a_s) and the base product to d->hid_ptr; NO task outputs (y_t / a_t are
                                 not touched) -- P (all segments) is written to ctx as usual */
#define MTLORA_HID_P_GIVEN 2  /* fwd (the Mlp's fc2): the TASK columns of P in ctx were written by mtlora_mlp_hid_proj before this call; the
                                 P pass runs for the shared source only, x_t is not read */
#define MTLORA_HID_Q_GIVEN 4  /* bwd (the Mlp's fc1): the TASK columns of Q (head of `scratch`) were written by mtlora_mlp_hid_bwd; dy_t is
                                 all-null, the dense part of dx comes from d->hid_ptr (G), dx_t / dA_t are formed from the given Q, dB_t
                                 must be null (mtlora_mlp_hid_bwd returns them) */

/* bytes of the context buffer written by fwd and read by bwd (packed low-rank factors + P; P alone when d->packed is set). */
int64_t mtlora_linear_ctx_bytes(const mtlora_linear_desc* d);

/* ---- packing the low-rank factors ahead of time (ABI v6).  fwd needs the fp32 masters A_* (r,K), B_* (N,r) of
 * models/lora.py:196-215 in the compute dtype and in several layouts (row-major, transposed, alpha-scaled projection rows,
 * fragment-major expansion factors): a small kernel per layer and call when done inside fwd.  The factors only change when the
 * optimizer steps, so a caller may keep one `packed` buffer per layer (mtlora_linear_packed_bytes) and refresh ALL of them with
 * one launch per step:
 *   mtlora_linear_pack            one layer, one launch (same arguments as fwd takes);
 *   mtlora_linear_pack_entry      writes one entry (mtlora_linear_pack_entry_bytes) of a HOST table describing a layer: master
 *                                 pointers, scales, geometry, destination; the caller copies the table to the device once (the
 *                                 pointers stay valid while parameters are updated in place);
 *   mtlora_linear_pack_table      ONE launch that packs every layer of a DEVICE table (entries of one dtype).
 * d->M is ignored by all of them; d->dropout_p / has_x_tasks / scales enter the alpha-scaled copies (train and eval differ). */
int64_t mtlora_linear_packed_bytes(const mtlora_linear_desc* d);
int mtlora_linear_pack(const mtlora_linear_desc* d, const float* A_s, const float* B_s, const float* const* A_t,
                       const float* const* B_t, void* packed, int64_t packed_bytes, void* stream);
int64_t mtlora_linear_pack_entry_bytes(void);
int mtlora_linear_pack_entry(const mtlora_linear_desc* d, const float* A_s, const float* B_s, const float* const* A_t,
                             const float* const* B_t, void* packed, int64_t packed_bytes, void* entry_host);
int mtlora_linear_pack_table(const void* table_dev, int n_entries, int dtype, void* stream);
/* bytes of the backward scratch buffer. */
int64_t mtlora_linear_bwd_scratch_bytes(const mtlora_linear_desc* d);

int mtlora_linear_fwd(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* W,
                      const float* bias, const float* A_s, const float* B_s, const float* const* A_t,
                      const float* const* B_t, void* y_s, void* const* y_t, void* ctx, int64_t ctx_bytes,
                      void* stream);

/* dy_s / dy_t[t] may be NULL (that output received no gradient).  dx_t is only written when
 * has_x_tasks; dA_x / dB_x may be NULL to skip a factor whose output got no gradient. */
int mtlora_linear_bwd(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* Wt,
                      const void* dy_s, const void* const* dy_t, const void* ctx, int64_t ctx_bytes, void* dx,
                      void* const* dx_t, float* dA_s, float* dB_s, float* const* dA_t, float* const* dB_t,
                      void* scratch, int64_t scratch_bytes, void* stream);

/* mtlora_linear_fwd that ALSO writes a_s = gelu(y_s), a_t[t] = gelu(y_t[t]) (exact erf form; same shape / dtype as the
 * outputs) from the output epilogue -- the Mlp's fc1 followed by its activation (swin_transformer_mtlora.py:57-78):
 * replaces the separate aten::gelu pass (read h, write a) by one extra write. */
int mtlora_linear_fwd_gelu(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* W,
                           const float* bias, const float* A_s, const float* B_s, const float* const* A_t,
                           const float* const* B_t, void* y_s, void* const* y_t, void* a_s, void* const* a_t, void* ctx,
                           int64_t ctx_bytes, void* stream);

/* mtlora_linear_bwd for a layer whose inputs are x = gelu(h_s), x_t[t] = gelu(h_t[t]) (the Mlp's fc2,
 * swin_transformer_mtlora.py:57-78: fc1 -> GELU -> fc2): dx and dx_t[t] are additionally multiplied by the exact-erf
 * GELU derivative gelu'(h) = Phi(h) + h phi(h) at the pre-activations (M x K, dtype of x), i.e. they are the gradients
 * w.r.t. h -- autograd's separate GeluBackward pass (read dX, read h, write dH) is folded into the dX epilogue.
 * h_t[t] is required wherever dx_t[t] is written. */
int mtlora_linear_bwd_gelu(const mtlora_linear_desc* d, const void* x, const void* const* x_t, const void* Wt,
                           const void* dy_s, const void* const* dy_t, const void* ctx, int64_t ctx_bytes, void* dx,
                           void* const* dx_t, float* dA_s, float* dB_s, float* const* dA_t, float* const* dB_t,
                           void* scratch, int64_t scratch_bytes, const void* h_s, const void* const* h_t, void* stream);

/* ------------------------------------------------------------------------------------------
 * Task-enabled Mlp with IMPLICIT task hidden tensors (ABI v8) -- replaces, for the T task streams of
 * `Mlp.forward` called with x_tasks (models/swin_transformer_mtlora.py:57-78: fc1 -> GELU -> fc2, both
 * MTLoRALinear with tasks, lora.py:262-266), the 3 T tensors of M x 4C elements the per-layer path writes
 * and re-reads (h_t = fc1 task outputs, gelu(h_t), and the gradients dH_t):
 *     h_t = h_base + P1_t B1_t^T   (h_base = x W1^T + b1, P1_t = s_t x_t A1_t^T: M x r_t)
 * enters fc2 ONLY through P2_t = s_t gelu(h_t) A2_t^T (M x r_t), and dH_t = (Q2_t A2_t) .* gelu'(h_t) is
 * consumed by G = dH_s + sum_t dH_t, Q1_t = s_t dH_t B1_t and the factor gradients dB1_t = dH_t^T P1_t,
 * dA2_t = Q2_t^T gelu(h_t).  Call sequence (d1 / d2 = descriptors of fc1 / fc2, same M, d1->N == d2->K):
 *   forward : mtlora_linear_fwd_gelu(d1 | HID_FWD_BASE) ; mtlora_mlp_hid_proj ; mtlora_linear_fwd(d2 | HID_P_GIVEN)
 *   backward: mtlora_linear_bwd_gelu(d2, dx_t = dA_t = null) ; mtlora_mlp_hid_bwd ; mtlora_linear_bwd(d1 | HID_Q_GIVEN)
 * Supported: 16-bit dtype, mode 'matrix', has_x_tasks, 1 <= T, every r_t <= 8, d1->N a multiple of 128; d1->N > 2048 only with
 * r_t <= 4 (the chunked MFMA kernels; mtlora_mlp_hid_supported returns 1).  Deterministic
 * (fixed-order partial sums).
 * ------------------------------------------------------------------------------------------ */
int mtlora_mlp_hid_supported(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2);
/* bytes of the scratch buffers of mtlora_mlp_hid_proj / mtlora_mlp_hid_bwd (-1: unsupported shape) */
int64_t mtlora_mlp_hid_fwd_scratch_bytes(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2);
int64_t mtlora_mlp_hid_bwd_scratch_bytes(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2);
/* ctx1: fc1's context (P1; and the packed factors when d1->packed is null); ctx2: fc2's context, whose task columns of P are written;
 * d2->packed must be set (mtlora_linear_pack / _pack_table before the call). */
int mtlora_mlp_hid_proj(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2, const void* h_base, const void* ctx1,
                        void* ctx2, void* scratch, int64_t scratch_bytes, void* stream);
/* dh_s: gradient w.r.t. the shared pre-activation (dx of fc2's bwd_gelu); scratch2: fc2's backward scratch after its phase 1 (holds
 * Q2); scratch1: fc1's backward scratch (its Q task columns are written); g: OUT (M x H) = dh_s + sum_t dH_t; dB1_t[t] (H x r_t) /
 * dA2_t[t] (r_t x H): fp32 OUT, nullable. */
int mtlora_mlp_hid_bwd(const mtlora_linear_desc* d1, const mtlora_linear_desc* d2, const void* h_base, const void* dh_s,
                       const void* ctx1, const void* ctx2, const void* scratch2, void* scratch1, void* g, float* const* dB1_t,
                       float* const* dA2_t, void* part, int64_t part_bytes, void* stream);

/* Weight gradient of a plain linear layer with a NARROW output (the decoder heads' final 1x1 convolutions,
 * models/seg_hrnet.py:518-526: nn.Conv2d(1080, num_classes, 1) on B*H*W pixels; autograd's dW = dY^T X):
 *   out (Na x Nb, fp32, row-major) = a^T b,  a = (M x Na, row stride lda), b = (M x Nb, row stride ldb).
 * Na, Nb, lda, ldb multiples of 8 (bf16) / 4 (f32); Na <= 1024.  Deterministic (fixed-order split-M partials). */
int64_t mtlora_gemm_tn_scratch_bytes(int64_t M, int Na, int Nb);
int mtlora_gemm_tn(const void* a, const void* b, float* out, int64_t M, int Na, int Nb, int64_t lda, int64_t ldb,
                   int dtype, void* scratch, int64_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Window attention core -- replaces swin_transformer_mtlora.py:194-220 (q*scale, q@k^T, + relative
 * position bias, + shift mask, softmax, @v, head merge) and, with image_layout = 1, also the
 * cyclic shift + window partition before it and the window merge + reverse shift after it
 * (:336-350, :365-377), by folding them into the kernel's load/store addressing.
 *
 * qkv: `dtype`, last dim laid out [3][num_heads][head_dim] (the reshape at :194-197).
 *   image_layout = 0: qkv is (n_windows, N, 3C) window-major exactly as the reference module sees it;
 *   image_layout = 1: qkv is (B, H, W, 3C) in natural token order; window w of image b covers
 *                     rows/cols ((wy*ws+ty+shift) mod H, (wx*ws+tx+shift) mod W).
 * out has the same token order as qkv, (.., C).  bias: dense (num_heads, N, N) fp32
 * (table[index] gathered by the host, :202-206).  mask: (nW_per_image, N, N) fp32 or NULL (:209-213).
 * N = ws*ws <= 64, head_dim == 32 (every Swin variant).
 * ------------------------------------------------------------------------------------------ */
typedef struct mtlora_attn_desc {
    int64_t B;            /* images */
    int32_t H, W;         /* token map */
    int32_t window_size;
    int32_t shift;        /* >= 0; only used for addressing when image_layout = 1 */
    int32_t num_heads;
    int32_t head_dim;
    int32_t image_layout;
    int32_t dtype;
    float scale;
    float mask_value;     /* value added where mask_ids differ (Swin: -100) */
} mtlora_attn_desc;

int64_t mtlora_window_attn_bwd_scratch_bytes(const mtlora_attn_desc* d);

/* bias: dense (num_heads, N, N) fp32, [h][i][j] added to score(query i, key j).  The shift mask can be given
 * either as region ids -- mask_ids (nW_per_image, N) int32: mask(i,j) = ids differ ? d->mask_value : 0, which is how
 * SW-MSA builds it (swin_transformer_mtlora.py:297-323); the fast path -- or as a general dense `mask`
 * (nW_per_image, N, N) fp32; both NULL = no mask.  When mask_ids is given, `mask` is ignored. */
int mtlora_window_attn_fwd(const mtlora_attn_desc* d, const void* qkv, const float* bias, const float* mask,
                           const int32_t* mask_ids, void* out, void* stream);
/* dbias: (num_heads, N, N) fp32 [h][i][j], overwritten.  dqkv: same shape/dtype as qkv, fully written. */
int mtlora_window_attn_bwd(const mtlora_attn_desc* d, const void* qkv, const float* bias, const float* mask,
                           const int32_t* mask_ids, const void* dout, void* dqkv, float* dbias, void* scratch,
                           int64_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (block glue around the path: norm1 / norm2 / PatchMerging.norm,
 * swin_transformer_mtlora.py:331-333, 395-396, 469) -- replaces aten::native_layer_norm + the autocast cast:
 * reads x (fp32 or bf16) once and writes y directly in the dtype the following linear consumes; fp32 statistics
 * (mean, rstd: (M) fp32) are saved for backward.  bwd also writes dgamma / dbeta (C) fp32 (overwritten).
 * merge_h, merge_w (0, 0 = off): PatchMerging (swin_transformer_mtlora.py:429-481) in front of its LayerNorm: x (and dx,
 * dx_addend) are the (B, merge_h*merge_w, C/4) token tensor and row (b, y2, x2) of the normalised (M, C) matrix is the
 * 2x2 neighbourhood [x(2y2,2x2) | x(2y2+1,2x2) | x(2y2,2x2+1) | x(2y2+1,2x2+1)] gathered on the fly (backward scatters).
 * dx_addend (nullable, dtype and shape of x): dx = dx_addend + LN-backward(dy).  In a transformer block x feeds the
 * LayerNorm AND the skip connection; passing the skip path's gradient here replaces the framework's separate
 * full-size gradient add.
 * ------------------------------------------------------------------------------------------ */
int mtlora_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t M, int64_t C, float eps, int x_dtype, int y_dtype, int merge_h, int merge_w, void* stream);
int64_t mtlora_layernorm_bwd_scratch_bytes(int64_t M, int64_t C, int x_dtype);
int mtlora_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         void* dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int x_dtype, int dy_dtype,
                         void* scratch, int64_t scratch_bytes, const void* dx_addend, int merge_h, int merge_w, void* stream);

/* Residual + DropPath fused with the LayerNorm that follows it (swin_transformer_mtlora.py:389-396: x = shortcut +
 * drop_path(attn(..)); then norm2(x); and :398-408 followed by the next block's norm1):
 *   fwd: x_new = shortcut + scale[sample] * branch  (written, dtype of shortcut = x_dtype);  y = LayerNorm(x_new) (y_dtype)
 *        branch has y_dtype; scale: (B) fp32 DropPath mask / keep, or NULL (= 1); rows M = B * tokens.
 *   bwd: d_shortcut = dx_addend + LN-backward(dy) (x_dtype);  d_branch = scale[sample] * d_shortcut (dy_dtype)
 * -- one pass each instead of residual kernel + LayerNorm kernel (saves re-reading the fp32 residual stream). */
int mtlora_residual_layernorm_fwd(const void* shortcut, const void* branch, const float* scale, int64_t B, const float* gamma,
                                  const float* beta, void* x_new, void* y, float* mean, float* rstd, int64_t M, int64_t C,
                                  float eps, int x_dtype, int y_dtype, void* stream);
int mtlora_residual_layernorm_bwd(const void* dy, const void* x_new, const float* gamma, const float* mean, const float* rstd,
                                  void* d_shortcut, void* d_branch, float* dgamma, float* dbeta, const float* scale,
                                  int64_t B, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                                  int64_t scratch_bytes, const void* dx_addend, void* stream);

/* n independent inputs through the SAME LayerNorm, one launch each way (PatchMerging.norm applied to the shared tensor and to
 * every task tensor, swin_transformer_mtlora.py:543-551); y[k] may be slices of one stacked buffer; dgamma / dbeta are summed
 * over the inputs; dx_addend (array, nullable entries) as in mtlora_layernorm_bwd. */
int64_t mtlora_layernorm_multi_bwd_scratch_bytes(int n, int64_t M, int64_t C, int x_dtype);
int mtlora_layernorm_multi_fwd(int n, const void* const* x, const float* gamma, const float* beta, void* const* y,
                               float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype, int y_dtype,
                               int merge_h, int merge_w, void* stream);
int mtlora_layernorm_multi_bwd(int n, const void* const* dy, const void* const* x, const float* gamma, const float* const* mean,
                               const float* const* rstd, void* const* dx, float* dgamma, float* dbeta, int64_t M, int64_t C,
                               int x_dtype, int dy_dtype, void* scratch, int64_t scratch_bytes, const void* const* dx_addend,
                               int merge_h, int merge_w, void* stream);

/* n independent streams, each  x_new[k] = res[k] + scale[k][sample] * branch[k]  followed by the SAME LayerNorm (plain rows, or
 * the PatchMerging gather with merge_h / merge_w): the MLP residual of the task-enabled block (swin_transformer_mtlora.py:398-408)
 * fused with the stage's PatchMerging norm over the shared + task tensors (:543-551).  res / branch / x_new / d_res / d_branch are
 * token tensors in the layout of x; y[k] may be slices of one stacked buffer.  Backward: d_res[k] = dx_addend[k] + LN-backward,
 * d_branch[k] = scale[k] * d_res[k], dgamma / dbeta summed over the streams (scratch: mtlora_layernorm_multi_bwd_scratch_bytes). */
int mtlora_residual_layernorm_streams_fwd(int n, const void* const* res, const void* const* branch, const float* scale,
                                          int64_t B, const float* gamma, const float* beta, void* const* x_new, void* const* y,
                                          float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype,
                                          int y_dtype, int merge_h, int merge_w, void* stream);
int mtlora_residual_layernorm_streams_bwd(int n, const void* const* dy, const void* const* x_new, const float* gamma,
                                          const float* const* mean, const float* const* rstd, void* const* d_res,
                                          void* const* d_branch, float* dgamma, float* dbeta, const float* scale, int64_t B,
                                          int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch, int64_t scratch_bytes,
                                          const void* const* dx_addend, int merge_h, int merge_w, void* stream);

/* The same for the task-enabled block (swin_transformer_mtlora.py:389-396 for the shared stream and each task stream): ONE
 * shortcut, n branches -> x_new[k] = shortcut + scale[k][sample] * branch[k], y[k] = LayerNorm(x_new[k]) (one launch), and
 * backward: d_branch[k] = scale[k] * (dx_addend[k] + LN-backward(dy[k])), d_shortcut = sum_k (dx_addend[k] + LN-backward(dy[k])),
 * dgamma / dbeta summed over the streams (one launch + reduce instead of n LayerNorm backwards and a residual backward).
 * scale: (n, B) fp32 or NULL; dx_addend[k], d_branch[k] nullable. */
int mtlora_residual_layernorm_multi_fwd(int n, const void* shortcut, const void* const* branch, const float* scale, int64_t B,
                                        const float* gamma, const float* beta, void* const* x_new, void* const* y,
                                        float* const* mean, float* const* rstd, int64_t M, int64_t C, float eps, int x_dtype,
                                        int y_dtype, void* stream);
int mtlora_residual_layernorm_multi_bwd(int n, const void* const* dy, const void* const* x_new, const float* gamma,
                                        const float* const* mean, const float* const* rstd, const void* const* dx_addend,
                                        void* d_shortcut, void* const* d_branch, float* dgamma, float* dbeta, const float* scale,
                                        int64_t B, int64_t M, int64_t C, int x_dtype, int dy_dtype, void* scratch,
                                        int64_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm + optional ReLU over a channels-last (R rows, C channels) matrix -- the decoder heads'
 * conv1x1 -> BatchNorm2d -> ReLU (seg_hrnet.py:498-526) on the (pixels, channels) matrix; replaces
 * aten::native_batch_norm(+backward) and the separate ReLU.  Biased variance for normalisation, unbiased for the
 * running_var update (momentum as nn.BatchNorm2d); running_* may be NULL.  save_* are (C) fp32 buffers written by
 * fwd and read by bwd (x is re-read by bwd; y is not needed).  dgamma / dbeta (C) fp32 are overwritten.
 * ------------------------------------------------------------------------------------------ */
int64_t mtlora_bn_scratch_bytes(int64_t R, int64_t C, int dtype);
int mtlora_bn_relu_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, int relu, void* y, float* save_mean, float* save_rstd,
                       float* save_scale, float* save_shift, int64_t R, int64_t C, int dtype, void* scratch,
                       int64_t scratch_bytes, void* stream);
int mtlora_bn_relu_bwd(const void* dy, const void* x, const float* save_mean, const float* save_rstd,
                       const float* save_scale, const float* save_shift, int relu, void* dx, float* dgamma, float* dbeta,
                       int64_t R, int64_t C, int dtype, void* scratch, int64_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * residual + DropPath for the n = 1+T tensors of a block half (swin_transformer_mtlora.py:389-392, 398-408):
 *   out_k = res_k + scale[k][sample] * y_k      (scale = DropPath mask / keep_prob, (n, B) fp32, NULL = ones)
 * (M, C) row-major, M = B * tokens, C % 8 == 0; res / out in res_dtype, y in y_dtype.  Backward: dy_k = scale * g_k
 * (y_dtype) and, when the residual is shared by every k, dres = sum_k g_k (res_dtype); g[k] may be NULL.
 * ------------------------------------------------------------------------------------------ */
int mtlora_residual_droppath_fwd(int n, const void* const* res, const void* const* y, void* const* out,
                                 const float* scale, int64_t M, int64_t C, int64_t B, int res_dtype, int y_dtype,
                                 void* stream);
int mtlora_residual_droppath_bwd(int n, const void* const* g, void* const* dy, void* dres, const float* scale,
                                 int64_t M, int64_t C, int64_t B, int res_dtype, int y_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss end of the train step, fused (callers): replaces the final
 * F.interpolate(pred, img_size, mode="bilinear") of models/swin_mtl.py:245 TOGETHER WITH the per-task loss of
 * mtl_loss_schemes.py and their autograd backward.
 *   kind 0  SoftMaxwithLoss (:22-39): cross entropy over C classes, ignore_index, mean over valid pixels;
 *           label (B,1,H,W) float class ids; stat[0] = number of valid pixels
 *   kind 1  NormalsLoss(normalize=True, L1, size_average) (:162-220): label (B,C,H,W), C <= 4;
 *           stat[0] = sum of the validity mask (label != ignore_index)
 *   kind 2  BalancedCrossEntropyLoss(size_average) (:42-89): C = 1, label (B,1,H,W);
 *           stat[0] = w = mean(1 - (label >= 0.5))
 * low (B,h,w,C) is the channels-last low-resolution prediction, H = scale*h, W = scale*w (integer scale 1..32,
 * align_corners=False; larger scales: MTLORA_ERR_UNSUPPORTED / a negative count).  Writes dlow = d loss / d low (dtype of
 * low) and `mtlora_upsample_loss_partials(B, h, w, scale)` fp32 partial loss values (one per tile of the launch) whose sum
 * is the loss.  `stat` is a device pointer (label-only statistics).  Deterministic: no float atomics.
 * ------------------------------------------------------------------------------------------ */
int64_t mtlora_upsample_loss_partials(int64_t B, int h, int w, int scale);
int mtlora_upsample_loss(int kind, const void* low, const float* label, const float* stat, void* dlow, float* partials,
                         int64_t B, int h, int w, int C, int scale, int dtype, float ignore_index, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small deterministic reductions of the callers (two launches each: per-block partials, fixed-order combine; no atomics and
 * no hipMemsetAsync, unlike ATen's multi-block reduce_kernel -- see csrc/reduce.hip and tools/find_memsets.py).
 *   mtlora_colsum      out[n] = sum_m x[m][n]  (fp32 out; x (M x N) row-major fp32 / bf16, N a multiple of the 16-byte vector):
 *                      replaces `grad.sum(0)` for the bias gradient of the heads' 1x1 convolutions
 *                      (reference models/seg_hrnet.py:498-526 layers under torch autograd)
 *   mtlora_label_stat  label-only statistics of the fused losses (`stat` of mtlora_upsample_loss), n fp32 labels:
 *                      kind 0: number of elements != ignore_index   (mtl_loss_schemes.py:22-39, :162-220)
 *                      kind 1: mean(1 - (label >= 0.5))             (mtl_loss_schemes.py:42-89)
 * ------------------------------------------------------------------------------------------ */
int64_t mtlora_colsum_scratch_bytes(int64_t M, int64_t N);
int mtlora_colsum(const void* x, int64_t M, int64_t N, int dtype, float* out, void* scratch, int64_t scratch_bytes, void* stream);
int64_t mtlora_label_stat_scratch_bytes(int64_t n);
int mtlora_label_stat(const float* label, int64_t n, int kind, float ignore_index, float* out, void* scratch,
                      int64_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Channels-last bilinear upsampling by an integer factor (align_corners=False) for the HRNet head of the callers
 * (models/seg_hrnet.py:498-526: F.interpolate of the coarse maps + torch.cat).  coarse (B,h,w,C) contiguous; the fine
 * tensor (B, scale*h, scale*w, .) is addressed with `ld_fine` elements per pixel, its pointer already offset to the
 * first channel of this map -- i.e. a channel slice of the concatenated matrix.  C and ld_fine multiples of 4,
 * pointers 8-byte (bf16) / 16-byte (fp32) aligned.  bwd = exact transpose (gather form, deterministic).
 * ------------------------------------------------------------------------------------------ */
int mtlora_upsample_cl_fwd(const void* coarse, void* fine, int64_t B, int h, int w, int C, int scale, int64_t ld_fine,
                           int dtype, void* stream);
int mtlora_upsample_cl_bwd(const void* grad_fine, void* grad_coarse, int64_t B, int h, int w, int C, int scale,
                           int64_t ld_fine, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * One whole SwinTransformerBlock WITHOUT task outputs, forward and backward, as ONE call each (ABI v7) -- replaces the
 * tasks-free path of SwinTransformerBlock.forward, models/swin_transformer_mtlora.py:326-408 (norm1 :331, shift / partition
 * :336-350, WindowAttention :353 = :186-227, merge / reverse shift :365-377, residual + DropPath :389, norm2 + Mlp :395-396
 * = :68-81, residual :398) with the four MTLoRALinear layers of lora.py:253-284 inside.  Nothing new is computed: the call
 * issues the launches of the entry points above in the block's order (LayerNorm -> qkv -> window attention -> proj ->
 * residual + DropPath + norm2 -> fc1 (+ GELU) -> fc2 -> residual + DropPath + the NEXT block's norm1), so its results are bit
 * identical to calling them one by one; what it removes is the caller's per-launch work (one host crossing and one autograd
 * node per block and direction instead of ~10: the train step of the reference's Swin-B config was bound by exactly that).
 * Every stage of a Swin backbone is `depth - 1` such blocks followed by one task-enabled block (:523-531), so a block of
 * this kind always hands over to another block: x_out AND normed_out = LayerNorm_next(x_out) are produced together.
 *
 * Tensors (M = B*H*W rows, token order (B, H*W) as everywhere in this library):
 *   x (M,C) x_dtype: the residual stream entering the block;  normed (M,C) dtype: norm1(x) when has_norm1 == 0 (it came out of
 *   the previous block's call), ignored otherwise;  x_out (M,C) x_dtype, normed_out (M,C) dtype.
 *   save: ONE caller-owned buffer of mtlora_block_save_bytes() that fwd fills and bwd reads (normalised tensors, qkv, attention
 *   output, the linears' ctx = P, the mid-block residual stream, fc1's pre-activation and activation, LayerNorm statistics);
 *   tmp: scratch of mtlora_block_fwd_tmp_bytes(), dead when fwd returns (stream-ordered).
 * bwd: g_x_out (M,C) x_dtype and g_normed_out (M,C) dtype are the gradients of the two outputs (g_x_out may be NULL);
 *   it writes g_x (x_dtype) and, when has_norm1 == 0, g_normed (dtype), the LayerNorm gradients, the eight factor gradients
 *   (fp32, shapes of the masters) and dbias (num_heads, N, N) fp32.  phase as mtlora_linear_desc.bwd_phase: 0 everything on
 *   `stream`; 1 everything but the factor gradients; 2 the factor gradients only (same arguments, a second stream ordered
 *   behind phase 1).  scratch: mtlora_block_bwd_scratch_bytes(), must stay valid until phase 2 has run.
 * ------------------------------------------------------------------------------------------ */
typedef struct mtlora_block_desc {
    int64_t B;
    int32_t H, W, C, hidden;   /* token map, channels, Mlp hidden width */
    int32_t num_heads, window_size, shift;
    int32_t dtype;             /* compute dtype: the linears, the normalised tensors, attention */
    int32_t x_dtype;           /* dtype of the residual stream (fp32 under autocast) */
    int32_t has_norm1;         /* 1: the call applies norm1 itself (first block of a stage); 0: `normed` is an input */
    float eps1, eps2, eps_next;
    float attn_scale, mask_value;
    mtlora_linear_desc lin[4]; /* qkv (C -> 3C), proj (C -> C), fc1 (C -> hidden, + GELU), fc2 (hidden -> C); T = 0, M = B*H*W;
                                  seed / dropout_p / packed / sel_* as for a stand-alone call (bwd_phase is set by the block call) */
} mtlora_block_desc;

typedef struct mtlora_block_params {   /* device pointers, caller-owned; fp32 unless noted */
    const float *norm1_g, *norm1_b;    /* has_norm1 only */
    const float *norm2_g, *norm2_b, *next_g, *next_b;
    const void* W[4];                  /* (N,K) dtype */
    const void* Wt[4];                 /* (K,N) dtype (bwd) */
    const float* bias[4];              /* (N) or NULL */
    const float* A[4];                 /* (r,K) masters */
    const float* Bf[4];                /* (N,r) masters */
    const float* attn_bias;            /* dense (num_heads, N, N) */
    const int32_t* mask_ids;           /* (nW, N) region ids of SW-MSA or NULL */
    const float* mask;                 /* general dense mask or NULL (ignored when mask_ids is given) */
    const float *scale1, *scale2;      /* DropPath mask / keep of the two residuals, (B) each, or NULL (= 1) */
} mtlora_block_params;

typedef struct mtlora_block_grads {    /* outputs of bwd, fp32 unless noted, all overwritten */
    void* g_x;                         /* (M,C) x_dtype */
    void* g_normed;                    /* (M,C) dtype; has_norm1 == 0 only */
    float *d_norm1_g, *d_norm1_b;      /* has_norm1 only */
    float *d_norm2_g, *d_norm2_b, *d_next_g, *d_next_b;
    float* dA[4];
    float* dB[4];
    float* dbias;                      /* (num_heads, N, N) */
} mtlora_block_grads;

int64_t mtlora_block_save_bytes(const mtlora_block_desc* d);
int64_t mtlora_block_fwd_tmp_bytes(const mtlora_block_desc* d);
int64_t mtlora_block_bwd_scratch_bytes(const mtlora_block_desc* d);
int mtlora_block_fwd(const mtlora_block_desc* d, const mtlora_block_params* p, const void* x, const void* normed, void* x_out,
                     void* normed_out, void* save, int64_t save_bytes, void* tmp, int64_t tmp_bytes, void* stream);
int mtlora_block_bwd(const mtlora_block_desc* d, const mtlora_block_params* p, const void* x, const void* normed,
                     const void* x_out, const void* g_x_out, const void* g_normed_out, const void* save, int64_t save_bytes,
                     const mtlora_block_grads* g, void* scratch, int64_t scratch_bytes, int phase, void* stream);

/* ------------------------------------------------------------------------------------------
 * Hardware self-test: writes the lane->element maps of the MFMA / LDS-transpose primitives the
 * kernels rely on into `out` (int32[4096]) so a GPU test can assert them (tests/test_gpu_layouts.py).
 * ------------------------------------------------------------------------------------------ */
int mtlora_selftest_layouts(int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Opt-in per-launch timing for the roofline report (bench.py): between begin and end every kernel
 * launch of this library is bracketed by two HIP events recorded on ITS launch stream and tagged with
 * a kind and the algorithmic bytes of that launch.  `end` must be called after the streams
 * were synchronised.  A process-wide diagnostic switch -- the only global state in the library; the
 * compute entry points stay re-entrant.
 * ------------------------------------------------------------------------------------------ */
#define MTLORA_PROF_KINDS 24
typedef struct mtlora_prof_summary {
    int64_t count[MTLORA_PROF_KINDS];
    double ms[MTLORA_PROF_KINDS];        /* sum of launch durations */
    double alg_bytes[MTLORA_PROF_KINDS]; /* sum of the useful bytes of the launches as issued (incl. the fused GELU write /
                                            gate read riding on the MTLoRALinear kernels) */
    double s8d_bytes[MTLORA_PROF_KINDS]; /* sum of the SURVEY 8(d) algorithmic bytes: the MTLoRALinear / attention-core formulas
                                            only (no GELU traffic); 0 for kinds outside the hot path (ABI v3) */
    double flops[MTLORA_PROF_KINDS];     /* sum of algorithmic FLOPs, un-padded ranks (GEMM kinds only; ABI v3) */
} mtlora_prof_summary;
int mtlora_prof_begin(int max_records);
int mtlora_prof_end(mtlora_prof_summary* out);
const char* mtlora_prof_kind_name(int kind);

#ifdef __cplusplus
}
#endif
#endif /* MTLORA_HIP_H */
