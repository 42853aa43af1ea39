// block.hip -- one whole SwinTransformerBlock without task outputs per call (include/mtlora_hip.h, ABI v7).
//
// Host code only: the call issues the launches of the library's own entry points (LayerNorm family, MTLoRALinear, window
// attention) in the order of SwinTransformerBlock.forward (reference models/swin_transformer_mtlora.py:326-408, tasks-free path)
// into caller-owned buffers.  Nothing is computed here that the stand-alone entry points do not compute -- the results are bit
// identical to calling them one by one (tests/test_gpu_models.py::test_block_call_matches_per_layer_calls); what disappears is
// the caller's work per launch: ~10 autograd nodes, ~100 view objects and ~20 ctypes marshals per block and direction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtlora_hip.h"
#include "internal.h"

namespace {

constexpr int64_t kAlign = 256;

inline int64_t es_of(int dtype) { return dtype == MTLORA_F32 ? 4 : 2; }
inline int64_t up(int64_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

// ---- the `save` buffer (written by fwd, read by bwd) ----
struct SaveLayout {
    int64_t xn, mean1, rstd1;          // has_norm1 only
    int64_t qkv, attn, ctx[4];         // qkv output, attention output (proj input), the four linears' ctx (= P [+ packed factors])
    int64_t x1, xn2, mean2, rstd2;     // residual stream after the attention half, norm2 of it
    int64_t h, a;                      // fc1 pre-activation, gelu(h)
    int64_t mean_n, rstd_n;            // statistics of the NEXT block's norm1 over x_out
    int64_t ctx_bytes[4];
    int64_t total;
};

// ---- backward scratch ----
struct BwdLayout {
    int64_t d_m, d_h, d_xn2, d_x1, d_y, d_a, d_qkv, d_xn, d_skip;  // activation gradients along the chain
    int64_t ln[3], attn, lin[4];                             // LayerNorm partials (next / norm2 / norm1: kept until phase 2 reduces them), attention, the linears' scratch (Q)
    int64_t ln_bytes, attn_bytes, lin_bytes[4];
    int64_t total;
};

bool desc_ok(const mtlora_block_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->hidden <= 0 || d->num_heads <= 0) return false;
    if (d->C % 8 || d->hidden % 8 || d->C % d->num_heads) return false;
    const int64_t M = d->B * d->H * d->W;
    const int64_t K[4] = {d->C, d->C, d->C, d->hidden}, N[4] = {3 * (int64_t)d->C, d->C, d->hidden, d->C};
    for (int i = 0; i < 4; ++i) {
        const mtlora_linear_desc& l = d->lin[i];
        if (l.M != M || l.K != K[i] || l.N != N[i] || l.T != 0 || l.dtype != d->dtype) return false;
    }
    return true;
}

int save_layout(const mtlora_block_desc* d, SaveLayout& L) {
    if (!desc_ok(d)) return MTLORA_ERR_SHAPE;
    const int64_t M = d->B * d->H * d->W, C = d->C, Hd = d->hidden, es = es_of(d->dtype), xs = es_of(d->x_dtype);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t at = o; o += up(bytes); return at; };
    L.xn = L.mean1 = L.rstd1 = -1;
    if (d->has_norm1) {
        L.xn = take(M * C * es);
        L.mean1 = take(M * 4);
        L.rstd1 = take(M * 4);
    }
    L.qkv = take(M * 3 * C * es);
    L.attn = take(M * C * es);
    for (int i = 0; i < 4; ++i) {
        L.ctx_bytes[i] = mtlora_linear_ctx_bytes(&d->lin[i]);
        if (L.ctx_bytes[i] < 0) return MTLORA_ERR_SHAPE;
        L.ctx[i] = take(L.ctx_bytes[i] > 16 ? L.ctx_bytes[i] : 16);
    }
    L.x1 = take(M * C * xs);
    L.xn2 = take(M * C * es);
    L.mean2 = take(M * 4);
    L.rstd2 = take(M * 4);
    L.h = take(M * Hd * es);
    L.a = take(M * Hd * es);
    L.mean_n = take(M * 4);
    L.rstd_n = take(M * 4);
    L.total = o;
    return MTLORA_OK;
}

mtlora_attn_desc attn_desc(const mtlora_block_desc* d) {
    mtlora_attn_desc a = {};
    a.B = d->B;
    a.H = d->H;
    a.W = d->W;
    a.window_size = d->window_size;
    a.shift = d->shift;
    a.num_heads = d->num_heads;
    a.head_dim = d->C / d->num_heads;
    a.image_layout = 1;
    a.dtype = d->dtype;
    a.scale = d->attn_scale;
    a.mask_value = d->mask_value;
    return a;
}

int bwd_layout(const mtlora_block_desc* d, BwdLayout& L) {
    if (!desc_ok(d)) return MTLORA_ERR_SHAPE;
    const int64_t M = d->B * d->H * d->W, C = d->C, Hd = d->hidden, es = es_of(d->dtype), xs = es_of(d->x_dtype);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t at = o; o += up(bytes); return at; };
    L.d_m = take(M * C * es);
    L.d_h = take(M * Hd * es);
    L.d_xn2 = take(M * C * es);
    L.d_x1 = take(M * C * xs);
    L.d_y = take(M * C * es);
    L.d_a = take(M * C * es);
    L.d_qkv = take(M * 3 * C * es);
    L.d_xn = d->has_norm1 ? take(M * C * es) : -1;
    L.d_skip = d->has_norm1 ? take(M * C * xs) : -1;  // skip-path gradient in front of norm1's backward
    L.ln_bytes = mtlora_layernorm_bwd_scratch_bytes(M, C, d->x_dtype);
    const mtlora_attn_desc a = attn_desc(d);
    L.attn_bytes = mtlora_window_attn_bwd_scratch_bytes(&a);
    if (L.ln_bytes < 0 || L.attn_bytes < 0) return MTLORA_ERR_SHAPE;
    for (int i = 0; i < 3; ++i) L.ln[i] = (i < 2 || d->has_norm1) ? take(L.ln_bytes) : -1;
    L.attn = take(L.attn_bytes > 16 ? L.attn_bytes : 16);
    for (int i = 0; i < 4; ++i) {
        L.lin_bytes[i] = mtlora_linear_bwd_scratch_bytes(&d->lin[i]);
        if (L.lin_bytes[i] < 0) return MTLORA_ERR_SHAPE;
        L.lin[i] = take(L.lin_bytes[i] > 16 ? L.lin_bytes[i] : 16);
    }
    L.total = o;
    return MTLORA_OK;
}

inline char* at(void* base, int64_t off) { return reinterpret_cast<char*>(base) + off; }
inline const char* at(const void* base, int64_t off) { return reinterpret_cast<const char*>(base) + off; }

#define BLK_TRY(call)                \
    do {                             \
        int st_ = (call);            \
        if (st_ != MTLORA_OK) return st_; \
    } while (0)

}  // namespace

extern "C" {

int64_t mtlora_block_save_bytes(const mtlora_block_desc* d) {
    SaveLayout L;
    return save_layout(d, L) == MTLORA_OK ? L.total : -1;
}

int64_t mtlora_block_fwd_tmp_bytes(const mtlora_block_desc* d) {
    if (!desc_ok(d)) return -1;
    return up(d->B * d->H * d->W * d->C * es_of(d->dtype));  // one (M, C) branch tensor: proj's output, then fc2's
}

int64_t mtlora_block_bwd_scratch_bytes(const mtlora_block_desc* d) {
    BwdLayout L;
    return bwd_layout(d, L) == MTLORA_OK ? L.total : -1;
}

int mtlora_block_fwd(const mtlora_block_desc* d, const mtlora_block_params* p, const void* x, const void* normed, void* x_out,
                     void* normed_out, void* save, int64_t save_bytes, void* tmp, int64_t tmp_bytes, void* stream) {
    SaveLayout L;
    BLK_TRY(save_layout(d, L));
    if (!p || !x || !x_out || !normed_out || !save || !tmp) return MTLORA_ERR_NULL;
    if (!d->has_norm1 && !normed) return MTLORA_ERR_NULL;
    if (save_bytes < L.total || tmp_bytes < mtlora_block_fwd_tmp_bytes(d)) return MTLORA_ERR_WORKSPACE;
    const int64_t M = d->B * d->H * d->W, C = d->C;
    const void* xn = normed;
    if (d->has_norm1) {  // norm1 (:331); the skip path keeps reading x itself
        if (!p->norm1_g || !p->norm1_b) return MTLORA_ERR_NULL;
        BLK_TRY(mtlora_layernorm_fwd(x, p->norm1_g, p->norm1_b, at(save, L.xn), (float*)at(save, L.mean1), (float*)at(save, L.rstd1), M,
                                     C, d->eps1, d->x_dtype, d->dtype, 0, 0, stream));
        xn = at(save, L.xn);
    }
    static const void* const kNoIn[MTLORA_MAX_TASKS] = {};
    static void* const kNoOut[MTLORA_MAX_TASKS] = {};
    static const float* const kNoF[MTLORA_MAX_TASKS] = {};
    // qkv on image-ordered tokens (:353 -> WindowAttention.forward :194), attention with the shift / partition / merge folded into
    // its addressing (:336-350, :365-377), proj (:222)
    BLK_TRY(mtlora_linear_fwd(&d->lin[0], xn, kNoIn, p->W[0], p->bias[0], p->A[0], p->Bf[0], kNoF, kNoF, at(save, L.qkv), kNoOut,
                              at(save, L.ctx[0]), L.ctx_bytes[0], stream));
    const mtlora_attn_desc ad = attn_desc(d);
    BLK_TRY(mtlora_window_attn_fwd(&ad, at(save, L.qkv), p->attn_bias, p->mask, p->mask_ids, at(save, L.attn), stream));
    BLK_TRY(mtlora_linear_fwd(&d->lin[1], at(save, L.attn), kNoIn, p->W[1], p->bias[1], p->A[1], p->Bf[1], kNoF, kNoF, tmp, kNoOut,
                              at(save, L.ctx[1]), L.ctx_bytes[1], stream));
    // x1 = x + DropPath(attn branch) and norm2(x1) in one pass (:389, :395)
    BLK_TRY(mtlora_residual_layernorm_fwd(x, tmp, p->scale1, d->B, p->norm2_g, p->norm2_b, at(save, L.x1), at(save, L.xn2),
                                          (float*)at(save, L.mean2), (float*)at(save, L.rstd2), M, C, d->eps2, d->x_dtype, d->dtype,
                                          stream));
    // Mlp (:68-81): fc1 with gelu(h) written by the same epilogue, fc2
    BLK_TRY(mtlora_linear_fwd_gelu(&d->lin[2], at(save, L.xn2), kNoIn, p->W[2], p->bias[2], p->A[2], p->Bf[2], kNoF, kNoF,
                                   at(save, L.h), kNoOut, at(save, L.a), kNoOut, at(save, L.ctx[2]), L.ctx_bytes[2], stream));
    BLK_TRY(mtlora_linear_fwd(&d->lin[3], at(save, L.a), kNoIn, p->W[3], p->bias[3], p->A[3], p->Bf[3], kNoF, kNoF, tmp, kNoOut,
                              at(save, L.ctx[3]), L.ctx_bytes[3], stream));
    // x_out = x1 + DropPath(mlp branch) (:398) and the next block's norm1 of it
    BLK_TRY(mtlora_residual_layernorm_fwd(at(save, L.x1), tmp, p->scale2, d->B, p->next_g, p->next_b, x_out, normed_out,
                                          (float*)at(save, L.mean_n), (float*)at(save, L.rstd_n), M, C, d->eps_next, d->x_dtype,
                                          d->dtype, stream));
    return MTLORA_OK;
}

int mtlora_block_bwd(const mtlora_block_desc* d, const mtlora_block_params* p, const void* x, const void* normed,
                     const void* x_out, const void* g_x_out, const void* g_normed_out, const void* save, int64_t save_bytes,
                     const mtlora_block_grads* g, void* scratch, int64_t scratch_bytes, int phase, void* stream) {
    SaveLayout L;
    BwdLayout S;
    BLK_TRY(save_layout(d, L));
    BLK_TRY(bwd_layout(d, S));
    if (!p || !g || !x || !x_out || !g_normed_out || !save || !scratch) return MTLORA_ERR_NULL;
    if (!d->has_norm1 && (!normed || !g->g_normed)) return MTLORA_ERR_NULL;
    if (save_bytes < L.total || scratch_bytes < S.total) return MTLORA_ERR_WORKSPACE;
    if (phase < 0 || phase > 2) return MTLORA_ERR_UNSUPPORTED;
    const int64_t M = d->B * d->H * d->W, C = d->C;
    const void* xn = d->has_norm1 ? (const void*)at(save, L.xn) : normed;
    static const void* const kNoIn[MTLORA_MAX_TASKS] = {};
    static void* const kNoOut[MTLORA_MAX_TASKS] = {};
    static float* const kNoG[MTLORA_MAX_TASKS] = {};
    mtlora_linear_desc ld[4];
    for (int i = 0; i < 4; ++i) {
        ld[i] = d->lin[i];
        ld[i].bwd_phase = phase;
    }
    // inputs / output gradients of the four linears (identical in both phases)
    const void* lin_x[4] = {xn, at(save, L.attn), at(save, L.xn2), at(save, L.a)};
    const void* lin_dy[4] = {at(scratch, S.d_qkv), at(scratch, S.d_y), at(scratch, S.d_h), at(scratch, S.d_m)};
    void* lin_dx[4] = {d->has_norm1 ? (void*)at(scratch, S.d_xn) : g->g_normed, at(scratch, S.d_a), at(scratch, S.d_xn2),
                       at(scratch, S.d_h)};
    auto linear_bwd = [&](int i) -> int {
        if (i == 3)  // fc2 reads a = gelu(h): its dX epilogue applies gelu'(h) and hands back the gradient w.r.t. h
            return mtlora_linear_bwd_gelu(&ld[3], lin_x[3], kNoIn, p->Wt[3], lin_dy[3], kNoIn, at(save, L.ctx[3]), L.ctx_bytes[3],
                                          lin_dx[3], kNoOut, g->dA[3], g->dB[3], kNoG, kNoG, at(scratch, S.lin[3]), S.lin_bytes[3],
                                          at(save, L.h), kNoIn, stream);
        return mtlora_linear_bwd(&ld[i], lin_x[i], kNoIn, p->Wt[i], lin_dy[i], kNoIn, at(save, L.ctx[i]), L.ctx_bytes[i], lin_dx[i],
                                 kNoOut, g->dA[i], g->dB[i], kNoG, kNoG, at(scratch, S.lin[i]), S.lin_bytes[i], stream);
    };
    const mtlora_attn_desc ad = attn_desc(d);
    auto ln_next = [&](int ph) -> int {      // next block's norm1 + the MLP residual: d_x1 = g_x_out + LN'(g_normed_out), d_m = scale2 * d_x1
        return mtli_residual_layernorm_bwd(g_normed_out, x_out, p->next_g, (const float*)at(save, L.mean_n),
                                           (const float*)at(save, L.rstd_n), at(scratch, S.d_x1), at(scratch, S.d_m), g->d_next_g,
                                           g->d_next_b, p->scale2, d->B, M, C, d->x_dtype, d->dtype, at(scratch, S.ln[0]), S.ln_bytes,
                                           g_x_out, ph, stream);
    };
    void* d_skip = d->has_norm1 ? (void*)at(scratch, S.d_skip) : g->g_x;
    auto ln_2 = [&](int ph) -> int {         // norm2 + the attention residual: d_x (skip part) = d_x1 + LN'(d_xn2), d_y = scale1 * that
        return mtli_residual_layernorm_bwd(at(scratch, S.d_xn2), at(save, L.x1), p->norm2_g, (const float*)at(save, L.mean2),
                                           (const float*)at(save, L.rstd2), d_skip, at(scratch, S.d_y), g->d_norm2_g, g->d_norm2_b,
                                           p->scale1, d->B, M, C, d->x_dtype, d->dtype, at(scratch, S.ln[1]), S.ln_bytes,
                                           at(scratch, S.d_x1), ph, stream);
    };
    auto ln_1 = [&](int ph) -> int {         // g_x = d_skip + LN1'(d_xn)
        return mtli_layernorm_bwd(at(scratch, S.d_xn), x, p->norm1_g, (const float*)at(save, L.mean1), (const float*)at(save, L.rstd1),
                                  g->g_x, g->d_norm1_g, g->d_norm1_b, M, C, d->x_dtype, d->dtype, at(scratch, S.ln[2]), S.ln_bytes,
                                  d_skip, ph, stream);
    };
    // (dbias is reduced on the chain's stream in every phase: the caller's gather table[index] -> bias has a backward of its own that
    // reads dbias as soon as this call returns)
    auto attn_bwd = [&]() -> int {
        return mtlora_window_attn_bwd(&ad, at(save, L.qkv), p->attn_bias, p->mask, p->mask_ids, at(scratch, S.d_a), at(scratch, S.d_qkv),
                                      g->dbias, at(scratch, S.attn), S.attn_bytes, stream);
    };
    if (!g->g_x || !g->d_norm2_g || !g->d_norm2_b || !g->d_next_g || !g->d_next_b || !g->dbias) return MTLORA_ERR_NULL;
    if (d->has_norm1 && (!p->norm1_g || !g->d_norm1_g || !g->d_norm1_b)) return MTLORA_ERR_NULL;
    if (phase == 2) {  // everything only the optimizer reads: the factor gradients of the four layers (in the order their inputs became
                       // available) and the second-stage reduces of the LayerNorm gradients
        BLK_TRY(ln_next(2));
        BLK_TRY(linear_bwd(3));
        BLK_TRY(linear_bwd(2));
        BLK_TRY(ln_2(2));
        BLK_TRY(linear_bwd(1));
        BLK_TRY(linear_bwd(0));
        if (d->has_norm1) BLK_TRY(ln_1(2));
        return MTLORA_OK;
    }
    BLK_TRY(ln_next(phase));
    BLK_TRY(linear_bwd(3));  // d_h
    BLK_TRY(linear_bwd(2));  // d_xn2
    BLK_TRY(ln_2(phase));
    BLK_TRY(linear_bwd(1));  // d_a
    BLK_TRY(attn_bwd());
    BLK_TRY(linear_bwd(0));  // d_xn (has_norm1) or g_normed
    if (d->has_norm1) BLK_TRY(ln_1(phase));
    return MTLORA_OK;
}

}  // extern "C"
