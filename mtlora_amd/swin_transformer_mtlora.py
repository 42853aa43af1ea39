"""MI355X drop-in for ``models.swin_transformer_mtlora`` (reference models/swin_transformer_mtlora.py).

Same public classes, constructor arguments, parameter / buffer names (so reference checkpoints and
``mark_only_lora_as_trainable`` name filters keep working) and the same tuple-returning calling
convention ``(shared, {task: tensor} | None)``; ``MultiTaskSwin`` (models/swin_mtl.py) and the decoder
heads run on top of it unchanged.

What is different is the dataflow of a block.  The reference rolls + partitions the normalised map,
runs qkv on window-ordered tokens, materialises (B_, nH, N, N) scores, and un-partitions + un-rolls
the result (and every per-task result) again (reference :326-408).  Here

  * qkv / proj / fc1 / fc2 are the fused MTLoRALinear kernels (csrc/linear.hip) and run on tokens in
    their NATURAL (B, H*W) order -- they are per-token maps, so the order is irrelevant to them;
  * the cyclic shift, window partition, bias + mask softmax attention, window merge and reverse shift
    are ONE kernel (csrc/attention.hip) that gathers the tokens of each shifted window by address
    and scatters the result back, so no permutation pass touches HBM and the per-task outputs of
    ``proj`` come out already in image order (the reference re-permutes each of them, :378-386).

``attention_layout = "windows"`` on a block restores the reference's window-ordered dataflow using
the stand-alone window-process kernels (csrc/window_process.hip) -- kept for parity tests and for
users of ``WindowAttention.forward(x_windows, mask)``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

import os

from . import functional as Fn
from .lora import MTLoRALinear
from .window_process import WindowProcess, WindowProcessReverse


# MTLORA_FUSED_BLOCKS=0: every block through the per-layer autograd Functions (A/B of the one-call path; same kernels, same results)
_FUSED_BLOCKS = os.environ.get("MTLORA_FUSED_BLOCKS", "1") != "0"


def set_fused_blocks(on: bool) -> bool:
    """switch the one-call-per-block path (functional.SwinBlockRunFn) on / off; returns the previous setting"""
    global _FUSED_BLOCKS
    prev, _FUSED_BLOCKS = _FUSED_BLOCKS, bool(on)
    return prev


# -- small stand-ins for the three timm helpers the reference imports (timm is not a dependency) ----
def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


trunc_normal_ = nn.init.trunc_normal_


class DropPath(nn.Module):
    """per-sample stochastic depth (timm DropPath semantics: scale kept samples by 1/keep)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            m.div_(keep)
        return x * m

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:.3f}"


class CompatLinear(nn.Linear):
    """plain linear with the tuple calling convention (reference :36-41); stays on rocBLAS/hipBLASLt."""

    def forward(self, input: Tensor, x_tasks: dict = None):
        if input.is_cuda and self.weight.requires_grad:  # huge-M trained linear (PatchMerging.reduction): split-K wgrad
            sh = input.shape
            y = Fn.linear_big_m(input.reshape(-1, sh[-1]), self.weight, self.bias)
            return y.view(*sh[:-1], y.shape[-1]), None
        return super().forward(input), None


def _mtlora_linear(mtlora, layer_idx, fin, fout, tasks, **kw):
    return MTLoRALinear(fin, fout, r=mtlora.R_PER_TASK_LIST[layer_idx], lora_shared_scale=mtlora.SHARED_SCALE[layer_idx],
                        lora_task_scale=mtlora.SCALE_PER_TASK_LIST[layer_idx], lora_dropout=mtlora.DROPOUT[layer_idx],
                        tasks=tasks, trainable_scale_shared=mtlora.TRAINABLE_SCALE_SHARED,
                        trainable_scale_per_task=mtlora.TRAINABLE_SCALE_PER_TASK, shared_mode=mtlora.SHARED_MODE, **kw)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, lora=False,
                 tasks=None, mtlora=None, layer_idx=0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        t = tasks if (lora or mtlora.INTERMEDIATE_SPECIALIZATION) else None
        self.fc1 = (_mtlora_linear(mtlora, layer_idx, in_features, hidden_features, t) if mtlora.FC1_ENABLED
                    else CompatLinear(in_features, hidden_features))
        self.act = act_layer()
        self.fc2 = (_mtlora_linear(mtlora, layer_idx, hidden_features, out_features, t) if mtlora.FC2_ENABLED
                    else CompatLinear(hidden_features, out_features))
        self.tasks = tasks
        self.drop = nn.Dropout(drop)

    def _fused_gelu_backward(self, h, h_t) -> bool:
        """fc2's dX kernel can apply the GELU derivative itself (one pass less over the 4C-wide hidden tensor per stream)."""
        from .lora import MTLoRALinear
        ts = [h] + ([h_t[t] for t in self.tasks] if h_t is not None else [])
        return (isinstance(self.fc2, MTLoRALinear) and type(self.act) is nn.GELU and getattr(self.act, "approximate", "none") == "none"
                and (self.drop.p == 0.0 or not self.training) and torch.is_grad_enabled() and h.requires_grad
                and all(t.is_cuda and t.is_contiguous() and t.dtype == Fn.compute_dtype(t) for t in ts)
                and self.fc2.linear.in_features % 8 == 0 and not self.fc2.linear.weight.requires_grad
                and self.fc2.shared_mode in ("matrix", "matrixv2"))

    def _fused_gelu(self, x) -> bool:
        """fc1 writes gelu(h) next to h from its epilogue and fc2's dX kernel applies gelu'(h): no separate GELU passes."""
        from .lora import MTLoRALinear
        ok = lambda m: (isinstance(m, MTLoRALinear) and m.shared_mode in ("matrix", "matrixv2") and not m.linear.weight.requires_grad
                        and m.linear.in_features % 8 == 0 and m.linear.out_features % 8 == 0)
        return (ok(self.fc1) and ok(self.fc2) and type(self.act) is nn.GELU and getattr(self.act, "approximate", "none") == "none"
                and (self.drop.p == 0.0 or not self.training) and torch.is_grad_enabled() and x.is_cuda and x.requires_grad
                and self.fc1.tasks == self.fc2.tasks)

    def _hid_forward(self, x, x_tasks):
        """task-enabled Mlp called with x_tasks: the T task hidden tensors (fc1's task outputs, their GELU, and the gradients w.r.t. them)
        stay implicit -- ``Fn.MlpHidFn`` (csrc/hid.hip); None when the call does not qualify (the per-layer path then runs)."""
        tasks = self.tasks
        if not (Fn.mlp_hid_enabled() and x_tasks is not None and tasks and self.fc1.tasks is not None
                and list(self.fc1.tasks) == list(tasks) and list(self.fc2.tasks) == list(tasks)
                and self.fc1.linear.out_features == self.fc2.linear.in_features and all(t in x_tasks for t in tasks)):
            return None
        dtype = Fn.compute_dtype(x)
        if dtype not in (torch.bfloat16, torch.float16) or x.shape[-1] != self.fc1.linear.in_features:
            return None
        xt = [x_tasks[t] for t in tasks]
        if (not x.is_cuda) or any(v.device != x.device or v.shape != x.shape for v in xt):
            return None
        # (shape rules first: ``hid_call`` draws the layers' dropout seeds)
        H, r1, r2 = self.fc1.linear.out_features, self.fc1._ranks, self.fc2._ranks
        if H % 128 or any(not (1 <= r1.get(t, 0) <= 8 and 1 <= r2.get(t, 0) <= 8) for t in tasks):  # (width limits: mlp_hid_supported)
            return None
        c1 = self.fc1.hid_call(dtype, x.device)
        c2 = self.fc2.hid_call(dtype, x.device) if c1 is not None else None
        if c1 is None or c2 is None:
            return None
        (m1, (W1c, W1t, b1), f1), (m2, (W2c, W2t, b2), f2) = c1, c2
        M = x.numel() // m1.K
        if not Fn.mlp_hid_supported(m1, m2, M):
            return None
        outs = Fn.MlpHidFn.apply(m1, m2, x, W1c, W1t, b1, W2c, W2t, b2, f1[0], f1[1], f2[0], f2[1], *xt, *f1[2], *f1[3], *f2[2], *f2[3])
        return outs[0], {t: outs[1 + i] for i, t in enumerate(tasks)}

    def forward(self, x, x_tasks=None):
        if x_tasks is not None and self._fused_gelu(x):
            out = self._hid_forward(x, x_tasks)
            if out is not None:
                return out
        if self._fused_gelu(x):
            h, h_t, a, a_t = self.fc1(x, x_tasks, gelu_out=True)
            return self.fc2(a, a_t, gelu_gate=(h.detach(), None if h_t is None else {t: v.detach() for t, v in h_t.items()}))
        h, h_t = self.fc1(x, x_tasks)
        if self._fused_gelu_backward(h, h_t):
            a = Fn.GeluDeferredGradFn.apply(h)
            a_t = {t: Fn.GeluDeferredGradFn.apply(h_t[t]) for t in self.tasks} if h_t is not None else None
            return self.fc2(a, a_t, gelu_gate=(h, h_t))
        h = self.drop(self.act(h))
        if h_t is not None:
            h_t = {t: self.drop(self.act(h_t[t])) for t in self.tasks}
        y, y_t = self.fc2(h, h_t)
        y = self.drop(y)
        if y_t is not None:
            y_t = {t: self.drop(y_t[t]) for t in self.tasks}
        return y, y_t


def window_partition(x, window_size):
    """(B,H,W,C) -> (nW*B, ws, ws, C) (pure-torch helper kept for API parity; reference :84-98)."""
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    """inverse of window_partition (reference :101-116)."""
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def _relative_position_index(wh: int, ww: int) -> Tensor:
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)])            # 2, N
    rel = pos[:, :, None] - pos[:, None, :]                        # 2, N, N
    return (rel[0] + wh - 1) * (2 * ww - 1) + (rel[1] + ww - 1)


class WindowAttention(nn.Module):
    """W-MSA / SW-MSA with relative position bias; qkv and proj are MTLoRALinear (reference :119-243)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 lora=False, tasks=None, mtlora=None, layer_idx=0):
        super().__init__()
        self.dim = dim
        self.window_size = window_size  # (Wh, Ww)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(*window_size))
        self.qkv = (_mtlora_linear(mtlora, layer_idx, dim, dim * 3, None, bias=qkv_bias) if mtlora.QKV_ENABLED
                    else CompatLinear(dim, dim * 3, bias=qkv_bias))
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = (_mtlora_linear(mtlora, layer_idx, dim, dim,
                                    tasks if (lora or mtlora.INTERMEDIATE_SPECIALIZATION) else None)
                     if mtlora.PROJ_ENABLED else CompatLinear(dim, dim))
        self.tasks = tasks
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.softmax = nn.Softmax(dim=-1)

    def dense_bias(self) -> Tensor:
        """(nH, N, N) = table[index] (reference :202-206); autograd routes the kernel's dbias into the table.
        On the GPU the gather is a product with a constant one-hot matrix (fp32, outside autocast): same values, and the
        backward is ONE deterministic GEMM  d table^T = dbias (nH x N^2) @ onehot^T  instead of index_put's sort +
        segmented-reduce kernels."""
        n = self.window_size[0] * self.window_size[1]
        table = self.relative_position_bias_table
        if not table.is_cuda:
            b = table[self.relative_position_index.view(-1)]
            return b.view(n, n, -1).permute(2, 0, 1).float()
        oh = getattr(self, "_rpi_onehot", None)
        if oh is None or oh.device != table.device:
            idx = self.relative_position_index.view(-1).to(table.device)
            oh = torch.zeros(table.shape[0], n * n, dtype=torch.float32, device=table.device)
            oh[idx, torch.arange(n * n, device=table.device)] = 1.0
            self._rpi_onehot = oh  # plain attribute: a constant derived from relative_position_index, not module state
        with torch.autocast("cuda", enabled=False):
            return (table.float().t() @ oh).view(-1, n, n)

    def _core(self, qkv: Tensor, meta: Fn.AttnMeta, mask: Optional[Tensor], mask_ids: Optional[Tensor]) -> Tensor:
        if self.training and self.attn_drop.p > 0:
            raise NotImplementedError("mtlora_amd: attn_drop > 0 is not supported by the fused attention kernel "
                                      "(every MTLoRA config uses 0)")
        return Fn.WindowAttentionFn.apply(meta, qkv, self.dense_bias(), mask, mask_ids)

    def _project(self, a: Tensor):
        y, y_t = self.proj(a)
        y = self.proj_drop(y)
        if y_t is not None:
            y_t = {t: self.proj_drop(y_t[t]) for t in self.tasks}
        return y, y_t

    def forward(self, x, mask=None):
        """x: (num_windows*B, N, C) window-ordered tokens, mask: (nW, N, N) or None -- the reference API (:186-227)."""
        B_, N, C = x.shape
        qkv, _ = self.qkv(x)
        nW = 1 if mask is None else mask.shape[0]
        ws = self.window_size[0]
        meta = Fn.AttnMeta(B=B_ // nW, H=ws, W=ws * nW, window_size=ws, shift=0, num_heads=self.num_heads,
                           head_dim=C // self.num_heads, image_layout=False, scale=self.scale)
        return self._project(self._core(qkv, meta, mask, None))

    def forward_image(self, x, H: int, W: int, shift: int, mask=None, mask_ids=None):
        """x: (B, H*W, C) tokens in image order.  Equivalent to roll(-shift) -> partition -> forward ->
        merge -> roll(+shift), with the permutations folded into the attention kernel's addressing."""
        B, L, C = x.shape
        qkv, _ = self.qkv(x)
        meta = Fn.AttnMeta(B=B, H=H, W=W, window_size=self.window_size[0], shift=shift, num_heads=self.num_heads,
                           head_dim=C // self.num_heads, image_layout=True, scale=self.scale)
        return self._project(self._core(qkv, meta, mask, mask_ids))

    def extra_repr(self) -> str:
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"

    def flops(self, N):
        return 4 * N * self.dim * self.dim + 2 * self.num_heads * N * N * (self.dim // self.num_heads)


def _shift_regions(H: int, W: int, ws: int, shift: int) -> Tensor:
    """(nW, N) region id (0..8) of every token of every shifted window (the ``img_mask`` of reference :300-314)."""
    ids = torch.zeros(H, W)
    edges_h = (0, H - ws, H - shift, H)
    edges_w = (0, W - ws, W - shift, W)
    for a in range(3):
        for b in range(3):
            ids[edges_h[a]:edges_h[a + 1], edges_w[b]:edges_w[b + 1]] = 3 * a + b
    return ids.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)


def _shift_mask(H: int, W: int, ws: int, shift: int) -> Tensor:
    """(nW, N, N) 0 / -100 mask of SW-MSA (reference :297-323): tokens of different cyclic regions do not attend."""
    win = _shift_regions(H, W, ws, shift)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 fused_window_process=False, lora=False, tasks=None, mtlora=None, layer_idx=0):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        self.tasks = tasks
        self.lora = lora
        if min(self.input_resolution) <= self.window_size:  # one window covers the map: no shift (reference :271-274)
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop, lora=lora,
                                    tasks=tasks, mtlora=mtlora, layer_idx=layer_idx)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop, lora=lora,
                       tasks=tasks, mtlora=mtlora, layer_idx=layer_idx)
        H, W = self.input_resolution
        mask = _shift_mask(H, W, self.window_size, self.shift_size) if self.shift_size > 0 else None
        self.register_buffer("attn_mask", mask)
        # the same mask as per-token region ids: what the attention kernel consumes (9 ids instead of N*N floats)
        self.register_buffer("_attn_mask_ids", None if mask is None else
                             _shift_regions(H, W, self.window_size, self.shift_size).to(torch.int32), persistent=False)
        self.fused_window_process = fused_window_process
        self.attention_layout = "image"  # "windows": reference dataflow through the window-process kernels

    # -- the whole block as ONE library call per direction (functional.SwinBlockRunFn, mtlora_block_fwd / _bwd) ------------------
    def _fusable_static(self, next_norm) -> bool:
        """STRUCTURE-only part of the eligibility test (cached per (block, next_norm) by ``BasicLayer``, dropped by its ``_apply`` and
        by ``invalidate_fused_cache``): the stock tasks-free block of every shipped config -- four MTLoRALinear layers with a shared
        'matrix' update, exact GELU, plain nn.LayerNorm's, 32-wide heads.  Nothing here can change without a module being swapped:
        what a ``.to()`` / ``.half()``, an attribute assignment or a hook CAN change (parameter dtypes / devices / contiguity,
        ``attention_layout``, the dropout probabilities, frozen / merged state) is looked at on every call in ``_block_call``."""
        def plain_ln(m):
            return type(m) is nn.LayerNorm and m.elementwise_affine and m.bias is not None and len(m.normalized_shape) == 1

        def stock_linear(m):
            return isinstance(m, MTLoRALinear) and m.r > 0 and m.tasks is None and hasattr(m, "lora_shared_A") and m.shared_mode == "matrix"

        lin = (self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2)
        C = self.dim
        return (not self.lora and next_norm is not None
                and all(stock_linear(m) for m in lin) and all(plain_ln(m) for m in (self.norm1, self.norm2, next_norm))
                and type(self.mlp.act) is nn.GELU and getattr(self.mlp.act, "approximate", "none") == "none"
                and all(isinstance(d, nn.Dropout) for d in (self.mlp.drop, self.attn.proj_drop, self.attn.attn_drop))
                and C % 8 == 0 and self.mlp.fc1.linear.out_features % 8 == 0
                and C // self.num_heads == 32 and self.window_size * self.window_size <= 64)

    def _block_call(self, has_norm1: bool, next_norm, cdtype, x):
        """(BlockCall, flat tensor list) of this block for a ``SwinBlockRunFn`` call, or None when the block cannot run fused NOW.
        Every eligibility check comes first; only then are the dropout seeds and DropPath factors drawn, in the order of ``forward``
        (a call that falls back must not have consumed seeds the per-layer path then draws again)."""
        lin = (self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2)
        dev = x.device
        # what can change between calls is looked at on every call (a few dozen attribute reads): a hook registered on any module
        # inside must see its module called, an un-frozen pretrained weight needs the dense gradients of the per-layer Function, a
        # merged weight has no shared update to apply, a model.to(bf16) / .half() leaves fp32-typed raw pointers stale
        if self.attention_layout != "image" or self.mlp.drop.p != 0.0 or self.attn.proj_drop.p != 0.0 or self.attn.attn_drop.p != 0.0:
            return None
        for m in (self, self.attn, self.mlp, self.norm1, self.norm2, next_norm, self.mlp.act, self.drop_path) + lin:
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                return None
        for n in (self.norm1, self.norm2, next_norm):
            for t in (n.weight, n.bias):
                if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                    return None
        for m in lin:
            if (m.merged or isinstance(m.lora_shared_scale, torch.Tensor) or m.rank_aware_shared() or m.linear.weight.requires_grad
                    or (m.linear.bias is not None and m.linear.bias.requires_grad)):
                return None
            for t in (m.lora_shared_A, m.lora_shared_B):
                if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                    return None
            if m.linear.weight.device != dev:
                return None
        tab = self.attn.relative_position_bias_table
        if tab.device != dev or (self._attn_mask_ids is not None and (self._attn_mask_ids.device != dev or self._attn_mask_ids.dtype != torch.int32
                                                                      or not self._attn_mask_ids.is_contiguous())):
            return None
        metas, weights, fparams = [], [], []
        for m in lin:
            metas.append(m.meta_t0(cdtype, dev))
            weights.append(m._weights(cdtype))
            fparams += [m.lora_shared_A, m.lora_shared_B]
        p_dp = self.drop_path.drop_prob if isinstance(self.drop_path, DropPath) else 0.0
        s1 = s2 = None
        if self.training and p_dp > 0.0:
            keep = 1.0 - p_dp
            s1 = Fn.droppath_scale(1, x.shape[0], keep, x.device)[0]
            s2 = Fn.droppath_scale(1, x.shape[0], keep, x.device)[0]
        H, W = self.input_resolution
        call = Fn.BlockCall(has_norm1, metas, weights, self.attn_mask, self._attn_mask_ids, H, W, self.num_heads, self.window_size,
                            self.shift_size, self.mlp.fc1.linear.out_features, (self.norm1.eps, self.norm2.eps, next_norm.eps),
                            self.attn.scale, tuple(fparams))
        flat = [self.attn.dense_bias(), s1, s2]
        if has_norm1:
            flat += [self.norm1.weight, self.norm1.bias]
        flat += [self.norm2.weight, self.norm2.bias, next_norm.weight, next_norm.bias]
        call.norm_params = tuple(flat[3:])
        flat += fparams
        return call, flat

    # -- attention half ------------------------------------------------------------------------------
    def _attend_windows(self, xn: Tensor, B: int, H: int, W: int, C: int):
        ws, s = self.window_size, self.shift_size
        xw = WindowProcess.apply(xn.view(B, H, W, C).contiguous(), B, H, W, C, -s, ws).view(-1, ws * ws, C)
        a, a_t = self.attn(xw, mask=self.attn_mask)
        merge = lambda v: WindowProcessReverse.apply(v.reshape(-1, ws, ws, C).contiguous(), B, H, W, C, s, ws).view(B, H * W, C)
        return merge(a), None if a_t is None else {t: merge(a_t[t]) for t in self.tasks}

    def forward(self, x, normed=None, next_norm=None, defer_residual=False):
        """normed: norm1(x) already formed by the previous block's fused residual + LayerNorm; next_norm: the following
        block's norm1 -- when this block ends in a single-stream residual the call returns (x, None, next_norm(x)).
        defer_residual: a task-enabled block returns (None, None, None, (residuals, branches, drop_prob)) instead of applying
        its MLP residual -- ``PatchMerging.forward_residual_multi`` fuses it into the merging norm."""
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        # x feeds norm1 AND the skip connection: one autograd node for both, so that the skip gradient is added inside the
        # LayerNorm backward kernel instead of by a separate full-size add (functional.LayerNormForkFn)
        if normed is None:
            shortcut, xn = Fn.layer_norm_fork(self.norm1, x)
        else:
            shortcut, xn = x, normed
        if self.attention_layout == "windows":
            a, a_t = self._attend_windows(xn, B, H, W, C)
        else:
            a, a_t = self.attn.forward_image(xn, H, W, self.shift_size, self.attn_mask, self._attn_mask_ids)
        # residuals: an independent DropPath draw per tensor, like the reference (:389-392); all 1+T of them in one
        # fused kernel (shared shortcut -> the backward also forms d_shortcut = sum of the 1+T gradients)
        p_dp = self.drop_path.drop_prob if isinstance(self.drop_path, DropPath) else 0.0
        x_t = None
        xn2_t = None
        if a_t is not None:  # shared shortcut, 1+T branches: residuals + DropPath + norm2 of every stream in one kernel
            xs, xns = Fn.residual_layer_norm_multi(self.norm2, shortcut, [a] + [a_t[t] for t in self.tasks], p_dp, self.training)
            x, xn2 = xs[0], xns[0]
            x_t = {t: xs[1 + i] for i, t in enumerate(self.tasks)}
            xn2_t = {t: xns[1 + i] for i, t in enumerate(self.tasks)}
        else:  # single stream: residual + DropPath + norm2 in one kernel
            x, xn2 = Fn.residual_layer_norm(self.norm2, shortcut, a, p_dp, self.training)
        # MLP half
        m, m_t = self.mlp(xn2, xn2_t)
        if m_t is None:
            if next_norm is not None:
                x, nxt = Fn.residual_layer_norm(next_norm, x, m, p_dp, self.training)
                return x, None, nxt
            return Fn.residual_droppath(x, [m], p_dp, self.training)[0], None
        if x_t is None:  # INTERMEDIATE_SPECIALIZATION-style: mlp specialises but attention did not (:401-403)
            out = Fn.residual_droppath(x, [m], p_dp, self.training)[0]
            return out, {t: self.drop_path(m_t[t]) for t in self.tasks}
        if defer_residual:  # the stage's PatchMerging applies x_k + DropPath(m_k) itself, inside its LayerNorm kernel
            return None, None, None, ([x] + [x_t[t] for t in self.tasks], [m] + [m_t[t] for t in self.tasks], p_dp)
        r = Fn.residual_droppath([x] + [x_t[t] for t in self.tasks], [m] + [m_t[t] for t in self.tasks], p_dp, self.training)
        return r[0], {t: r[1 + i] for i, t in enumerate(self.tasks)}

    def extra_repr(self) -> str:
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")

    def flops(self):
        H, W = self.input_resolution
        nW = H * W / self.window_size / self.window_size
        return (2 * self.dim * H * W + nW * self.attn.flops(self.window_size * self.window_size)
                + 2 * H * W * self.dim * self.dim * self.mlp_ratio)


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat -> LayerNorm(4C) -> reduction 4C->2C (reference :429-481)."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm, layer_idx=0, mtlora=None):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.reduction = (_mtlora_linear(mtlora, layer_idx, 4 * dim, 2 * dim, None, bias=False)
                          if mtlora.DOWNSAMPLER_ENABLED else CompatLinear(4 * dim, 2 * dim, bias=False))
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."
        # rows of the normalised matrix = 2x2 neighbourhoods, channel order [x(0,0), x(1,0), x(0,1), x(1,1)] as the
        # reference's cat; gathered by the LayerNorm kernel itself (no strided copy forward, no scatter copy backward)
        y, _ = self.reduction(Fn.layer_norm_merge(self.norm, x, H, W))
        return y

    def forward_residual_multi(self, res, branches, drop_prob, training):
        """merging of (res_k + DropPath(branches_k)) for the shared + task streams: residual, 2x2 gather and LayerNorm in one
        launch, one reduction GEMM; falls back to residual kernel + ``forward_multi``."""
        H, W = self.input_resolution
        stacked = Fn.residual_merge_norm_streams(self.norm, res, branches, H, W, drop_prob, training)
        if stacked is None:
            return self.forward_multi(Fn.residual_droppath(list(res), list(branches), drop_prob, training))
        y, _ = self.reduction(stacked)
        return list(y.view(len(res), -1, *y.shape[1:]).unbind(0))

    def forward_multi(self, xs):
        """the shared tensor and the task tensors through the same merging in ONE LayerNorm launch and ONE reduction GEMM
        (stacked along the batch); falls back to per-tensor calls when the fused kernel does not apply."""
        H, W = self.input_resolution
        stacked = Fn.layer_norm_merge_multi(self.norm, xs, H, W) if len(xs) > 1 else None
        if stacked is None:
            return [self.forward(x) for x in xs]
        y, _ = self.reduction(stacked)
        return list(y.view(len(xs), -1, *y.shape[1:]).unbind(0))

    def extra_repr(self) -> str:
        return f"input_resolution={self.input_resolution}, dim={self.dim}"

    def flops(self):
        H, W = self.input_resolution
        return H * W * self.dim + (H // 2) * (W // 2) * 4 * self.dim * 2 * self.dim


class BasicLayer(nn.Module):
    """one stage: `depth` blocks (odd ones shifted; only the last is task-specialised) + PatchMerging that is
    applied to the shared tensor and to every task tensor (reference :484-562)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 fused_window_process=False, tasks=None, mtlora=None, layer_idx=0):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.tasks = tasks
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if i % 2 == 0 else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer, fused_window_process=fused_window_process,
                                 lora=(i == depth - 1), tasks=tasks, mtlora=mtlora, layer_idx=layer_idx)
            for i in range(depth)])
        self.downsample = (downsample(input_resolution, dim=dim, norm_layer=norm_layer, layer_idx=layer_idx, mtlora=mtlora)
                           if downsample is not None else None)

    def _fused_run(self, x):
        """the leading run of blocks that can go through ``Fn.SwinBlockRunFn`` (one autograd node for the run, one library call per
        block and direction): (number of blocks consumed, x, normed) -- (0, x, None) when the first block already does not qualify."""
        ok = (_FUSED_BLOCKS and x.is_cuda and x.dim() == 3 and x.is_contiguous() and torch.is_grad_enabled() and x.dtype in Fn._GLUE_DTYPES)
        if not ok:
            return 0, x, None
        cdtype = Fn.compute_dtype(x)
        C = x.shape[-1]
        if C > (2048 if x.dtype == torch.float32 else 4096) or (x.dtype != torch.float32 and x.dtype != cdtype):
            return 0, x, None
        st = self.__dict__.get("_fusable_cache")
        if st is None:
            st = [blk._fusable_static(self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None) for i, blk in enumerate(self.blocks)]
            self.__dict__["_fusable_cache"] = st
        calls, flat = [], []
        for i, blk in enumerate(self.blocks):
            if not st[i] or blk.training != self.training:
                break
            cf = blk._block_call(i == 0, self.blocks[i + 1].norm1, cdtype, x)
            if cf is None:
                break
            calls.append(cf[0])
            flat += cf[1]
        if not calls:
            return 0, x, None
        x, normed = Fn.SwinBlockRunFn.apply(calls, x, None, *flat)
        return len(calls), x, normed

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .half(): nothing cached about the blocks survives a conversion
        self.__dict__.pop("_fusable_cache", None)
        return super()._apply(fn, *a, **k)

    def invalidate_fused_cache(self) -> None:
        """forget which blocks qualify STRUCTURALLY for the one-call path (after swapping modules inside a block; hooks, frozen / merged
        state are looked at on every call and need nothing)"""
        self.__dict__.pop("_fusable_cache", None)

    def forward(self, x):
        tasks_lora = None
        deferred = None
        first, x, normed = self._fused_run(x)
        for i, blk in enumerate(self.blocks):
            if i < first:
                continue
            # a block that ends in a single-stream residual also applies the next block's norm1 (one fused kernel)
            nxt = self.blocks[i + 1].norm1 if (i + 1 < len(self.blocks) and not blk.lora) else None
            last = i + 1 == len(self.blocks)
            defer = (last and blk.lora and self.downsample is not None and hasattr(self.downsample, "forward_residual_multi")
                     and self.training and torch.is_grad_enabled())
            out = blk(x, normed, nxt, defer) if defer else blk(x, normed, nxt)
            x, tasks_lora = out[0], out[1]
            normed = out[2] if len(out) > 2 else None
            deferred = out[3] if len(out) > 3 else None
        if deferred is not None:
            outs = self.downsample.forward_residual_multi(deferred[0], deferred[1], deferred[2], self.training)
            return outs[0], {t: outs[1 + i] for i, t in enumerate(self.tasks)}
        if self.downsample is not None:
            if tasks_lora is not None and hasattr(self.downsample, "forward_multi"):
                outs = self.downsample.forward_multi([x] + [tasks_lora[t] for t in self.tasks])
                x, tasks_lora = outs[0], {t: outs[1 + i] for i, t in enumerate(self.tasks)}
            else:
                x = self.downsample(x)
                if tasks_lora is not None:
                    tasks_lora = {t: self.downsample(tasks_lora[t]) for t in self.tasks}
        return x, tasks_lora

    def extra_repr(self) -> str:
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"

    def flops(self):
        return sum(b.flops() for b in self.blocks) + (self.downsample.flops() if self.downsample is not None else 0)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        # a kernel==stride convolution is a per-patch linear map: gather the (C, ph, pw) patches and run one GEMM
        # (same `proj.weight` / `proj.bias` parameters and values as the reference's Conv2d, no MIOpen find)
        ph, pw = self.patch_size
        p = x.view(B, C, H // ph, ph, W // pw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, -1, C * ph * pw)
        sh = p.shape
        x = Fn.linear_big_m(p.reshape(-1, sh[-1]), self.proj.weight.view(self.embed_dim, -1), self.proj.bias)
        x = x.view(*sh[:-1], self.embed_dim)
        return x if self.norm is None else Fn.layer_norm(self.norm, x, feeds_linear=False)

    def flops(self):
        Ho, Wo = self.patches_resolution
        f = Ho * Wo * self.embed_dim * self.in_chans * (self.patch_size[0] * self.patch_size[1])
        return f + (Ho * Wo * self.embed_dim if self.norm is not None else 0)


class SwinTransformerMTLoRA(nn.Module):
    """Swin backbone whose last block per stage emits one feature map per task (reference :616-772)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, fused_window_process=False, basic_layer=BasicLayer, tasks=None, mtlora=None,
                 **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = ape
        self.patch_norm = patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.tasks = tasks
        self.mtlora = mtlora
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        res = self.patch_embed.patches_resolution
        self.patches_resolution = res
        if ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            trunc_normal_(self.absolute_pos_embed, std=0.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]  # stochastic depth decay rule
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(basic_layer(
                dim=int(embed_dim * 2 ** i), input_resolution=(res[0] // 2 ** i, res[1] // 2 ** i), depth=depths[i],
                num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint,
                fused_window_process=fused_window_process, tasks=tasks, mtlora=mtlora, layer_idx=i))
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    def forward_features(self, x, return_stages=False, flatten_ft=False):
        x = self.patch_embed(x)
        if self.ape:
            x = x + self.absolute_pos_embed
        x = self.pos_drop(x)
        stages = []
        for layer in self.layers:
            x, tasks_lora = layer(x)
            if tasks_lora is None:
                tasks_lora = {t: x for t in self.tasks}
            stages.append((x, tasks_lora))
        if return_stages:
            return stages
        if flatten_ft:
            x = torch.flatten(self.avgpool(x.transpose(1, 2)), 1)
        return x

    def forward(self, x, return_stages=False, flatten_ft=False):
        return self.head(self.forward_features(x, return_stages, flatten_ft))

    def flops(self, images=None, logger=None, detailed=False):
        f = self.patch_embed.flops() + sum(l.flops() for l in self.layers)
        f += self.num_features * self.patches_resolution[0] * self.patches_resolution[1] // (2 ** self.num_layers)
        return f + self.num_features * self.num_classes
