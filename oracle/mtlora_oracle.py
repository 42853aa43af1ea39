"""CPU oracle for the MTLoRA hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (``mtlora_amd``) never
does; it fails loudly when the HIP library is missing.

This is a plain-PyTorch (fp32 / fp64, CPU-capable) *restatement* of the
reference algorithm, written in a functional style: every function takes the
tensors it needs (or a flat ``{state_dict_name: tensor}`` mapping that uses the
reference's parameter names) and returns tensors.  No code is shared with
``mtlora_amd`` and nothing here touches ``/root/reference``.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the real
reference modules (``/root/reference/models/{lora,swin_transformer_mtlora,
swin_mtl}.py`` and ``mtl_loss_schemes.py``) in the build container, ran them
on seeded inputs and committed inputs + outputs + gradients under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every function below
against those vectors (the reference's own test-suite pins nothing on this path
except the window-process index math, unit_test.py:96-115, which the golden
window fixtures reproduce).

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# counter-based dropout mask shared by the HIP kernels, the oracle and the
# tests.  NOT part of the reference (it uses torch's Philox stream, which no
# other implementation can reproduce); the *distribution* is what the
# reference specifies (nn.Dropout(p): keep w.p. 1-p, scale kept by 1/(1-p),
# lora.py:79-82, 258).  The generator is specified here so that the GPU kernel
# can be checked element-for-element.
# --------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32-style integer finaliser on uint32 arrays (wraps mod 2**32)."""
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(16))) & _M32
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x = (x ^ (x >> np.uint64(15))) & _M32
    x = (x * np.uint64(0x846CA68B)) & _M32
    x = (x ^ (x >> np.uint64(16))) & _M32
    return x


def dropout_keep_mask(seed: int, stream: int, rows: int, cols: int, p: float) -> Tensor:
    """keep[m, k] for element (m, k) of an (rows, cols) activation.

        rh   = mix32(m * 0x9E3779B1 + seed_lo + stream * 0x85EBCA77)          (row hash)
        h    = mix32(rh ^ ((k >> 1) + seed_hi * 0x27D4EB2F))                    (32 bits per element PAIR)
        bits = h >> 16 if k odd else h & 0xFFFF
        keep <=> bits >= floor(p * 65536)
    Mirrors ``mtl_dropout_keep`` in mtlora_amd/csrc/common.h.
    """
    if p <= 0.0:
        return torch.ones(rows, cols, dtype=torch.bool)
    seed_lo = np.uint64(seed & 0xFFFFFFFF)
    seed_hi = np.uint64((seed >> 32) & 0xFFFFFFFF)
    m = np.arange(rows, dtype=np.uint64)[:, None]
    k = np.arange(cols, dtype=np.uint64)[None, :]
    rh = _mix32((m * np.uint64(0x9E3779B1) + seed_lo + np.uint64(stream) * np.uint64(0x85EBCA77)) & _M32)
    h = _mix32((rh ^ (((k >> np.uint64(1)) + seed_hi * np.uint64(0x27D4EB2F)) & _M32)) & _M32)
    bits = np.where((k & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    thr = np.uint64(min(int(p * 65536.0), 65535))
    return torch.from_numpy(bits >= thr)


def dropout_keep_mask_t(seed: int, stream: int, rows: int, cols: int, p: float, device="cpu", row0: int = 0) -> Tensor:
    """:func:`dropout_keep_mask` restated with torch int64 arithmetic (any device; rows ``row0 .. row0 + rows``): the full-size
    GPU parity tests need the mask of 4e5 x 384 activations, which the numpy form builds in minutes.  Pinned to the numpy
    form element for element by tests/test_oracle_golden.py."""
    if p <= 0.0:
        return torch.ones(rows, cols, dtype=torch.bool, device=device)
    M32 = 0xFFFFFFFF

    def mix(x):  # x in [0, 2^32): int64 products wrap mod 2^64, the low 32 bits are exact
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & M32
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & M32
        return x ^ (x >> 16)

    seed_lo, seed_hi = seed & M32, (seed >> 32) & M32
    m = torch.arange(row0, row0 + rows, dtype=torch.int64, device=device)[:, None]
    k = torch.arange(cols, dtype=torch.int64, device=device)[None, :]
    rh = mix((m * 0x9E3779B1 + seed_lo + stream * 0x85EBCA77) & M32)
    h = mix(rh ^ (((k >> 1) + seed_hi * 0x27D4EB2F) & M32))
    bits = torch.where((k & 1) == 1, h >> 16, h & 0xFFFF)
    return bits >= min(int(p * 65536.0), 65535)


# --------------------------------------------------------------------------
# a2/a3/a4  MTLoRALinear  (models/lora.py:159-284)
# --------------------------------------------------------------------------
def mtlora_linear(
    x: Tensor,
    W: Tensor,
    b: Optional[Tensor],
    A_s: Optional[Tensor],
    B_s: Optional[Tensor],
    scale_s: float,
    tasks: Optional[Sequence[str]] = None,
    A_t: Optional[Mapping[str, Tensor]] = None,
    B_t: Optional[Mapping[str, Tensor]] = None,
    scale_t: Optional[Mapping[str, float]] = None,
    x_tasks: Optional[Mapping[str, Tensor]] = None,
    shared_mode: str = "matrix",
    keep_mask: Optional[Tensor] = None,
    p: float = 0.0,
    lora_norm: Optional[Tuple[Tensor, Tensor]] = None,
    torch_dropout: bool = False,
) -> Tuple[Tensor, Optional[Dict[str, Tensor]]]:
    """lora.py:253-284.  ``keep_mask`` (same shape as x, bool) stands in for
    ``self.lora_dropout`` (lora.py:258: the dropped x is re-bound, so the task
    path sees it too when ``x_tasks`` is None); ``torch_dropout`` applies
    ``F.dropout(x, p)`` itself (what nn.Dropout does; used by the eager-GPU baseline)."""
    pretrained = F.linear(x, W, b)                                   # :255
    if A_s is None and not tasks:                                    # r == 0 -> :256
        return pretrained, None
    xd = x
    if torch_dropout and p > 0.0:
        xd = F.dropout(x, p, training=True)
    elif keep_mask is not None and p > 0.0:
        xd = x * keep_mask.to(x.dtype) / (1.0 - p)                   # :258 nn.Dropout semantics

    def task_out(t, base):
        xin = xd if x_tasks is None else x_tasks[t]
        return base + (xin @ A_t[t].transpose(0, 1) @ B_t[t].transpose(0, 1)) * scale_t[t]

    if shared_mode == "matrix":                                      # :259-266
        lora = (xd @ A_s.transpose(0, 1) @ B_s.transpose(0, 1)) * scale_s
        lt = {t: task_out(t, pretrained) for t in tasks} if tasks else None
    elif shared_mode == "matrixv2":                                  # :267-274
        lora = (xd @ A_s.transpose(0, 1) @ B_s.transpose(0, 1)) * scale_s
        lt = {t: task_out(t, pretrained + lora) for t in tasks} if tasks else None
    elif shared_mode == "addition":                                  # :275-282
        lt = {t: task_out(t, pretrained) for t in tasks}
        s = torch.stack(list(lt.values()), 0).sum(0)
        lora = F.layer_norm(s, (s.shape[-1],), lora_norm[0], lora_norm[1])
    else:
        raise NotImplementedError(shared_mode)
    return pretrained + lora, lt                                     # :284


def mtlora_linear_backward_closed_form(
    x, W, A_s, B_s, scale_s, tasks, A_t, B_t, scale_t, x_tasks, dy_s, dy_t
):
    """Closed-form backward of the 'matrix' mode with no dropout.
    Used to cross-check autograd of :func:`mtlora_linear` and as the statement
    of what the HIP backward must produce."""
    G = dy_s.clone()
    for t in tasks or []:
        G = G + dy_t[t]
    out = {}
    Qs = (dy_s @ B_s) * scale_s                      # (M, r_s)
    dx = G @ W + Qs @ A_s
    out["dA_s"] = Qs.transpose(0, 1) @ x
    out["dB_s"] = (dy_s.transpose(0, 1) @ (x @ A_s.transpose(0, 1))) * scale_s
    out["dx_t"] = {}
    for t in tasks or []:
        xin = x if x_tasks is None else x_tasks[t]
        Qt = (dy_t[t] @ B_t[t]) * scale_t[t]
        if x_tasks is None:
            dx = dx + Qt @ A_t[t]
        else:
            out["dx_t"][t] = Qt @ A_t[t]
        out[f"dA_t.{t}"] = Qt.transpose(0, 1) @ xin
        out[f"dB_t.{t}"] = (dy_t[t].transpose(0, 1) @ (xin @ A_t[t].transpose(0, 1))) * scale_t[t]
    out["dx"] = dx
    return out


# --------------------------------------------------------------------------
# a7/a8  window partition / reverse / cyclic shift
# (swin_transformer_mtlora.py:84-116, 336-386; kernels/window_process/*.cu;
#  oracle of the CUDA kernels = unit_test.py:96-115)
# --------------------------------------------------------------------------
def window_partition(x: Tensor, ws: int) -> Tensor:
    """(B,H,W,C) -> (B*nH*nW, ws, ws, C), batch-major then row-major windows (:84-98)."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def window_reverse(win: Tensor, ws: int, H: int, W: int) -> Tensor:
    """inverse of :func:`window_partition` (:101-116)."""
    nH, nW = H // ws, W // ws
    B = win.shape[0] // (nH * nW)
    x = win.reshape(B, nH, nW, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def roll_and_window_partition(x: Tensor, shift: int, ws: int) -> Tensor:
    """window_partition(torch.roll(x, (-shift,-shift), (1,2)))  (block fwd :336-350;
    WindowProcess.apply(x,B,H,W,C,-shift,ws), window_process.py:13)."""
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    return window_partition(x, ws)


def window_merge_and_roll(win: Tensor, shift: int, ws: int, H: int, W: int) -> Tensor:
    """torch.roll(window_reverse(w), (shift,shift), (1,2))  (block fwd :365-377)."""
    x = window_reverse(win, ws, H, W)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    return x


# --------------------------------------------------------------------------
# a9  WindowAttention  (swin_transformer_mtlora.py:143-162, 186-227, 297-323)
# --------------------------------------------------------------------------
def relative_position_index(ws: int) -> Tensor:
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table (:147-162)."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shifted_window_mask(H: int, W: int, ws: int, shift: int) -> Optional[Tensor]:
    """(nW, ws*ws, ws*ws) with 0 / -100 (:297-323); None when shift == 0."""
    if shift <= 0:
        return None
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def dense_relative_bias(table: Tensor, ws: int) -> Tensor:
    """(nH, ws*ws, ws*ws) from the ((2ws-1)^2, nH) table (:202-206)."""
    n = ws * ws
    idx = relative_position_index(ws).reshape(-1).to(table.device)
    return table[idx].reshape(n, n, -1).permute(2, 0, 1).contiguous()


def window_attention_core(qkv: Tensor, bias: Tensor, mask: Optional[Tensor], num_heads: int,
                          scale: Optional[float] = None) -> Tensor:
    """qkv (B_, N, 3C) laid out [3][nH][hd] on the last dim -> (B_, N, C)  (:194-220).
    ``bias`` is the dense (nH, N, N) relative-position bias, ``mask`` (nW, N, N) or None."""
    B_, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // num_heads
    scale = hd ** -0.5 if scale is None else scale
    q, k, v = qkv.reshape(B_, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.reshape(B_ // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.reshape(-1, num_heads, N, N)
    attn = attn.softmax(-1)
    return (attn @ v).transpose(1, 2).reshape(B_, N, C)


# --------------------------------------------------------------------------
# parameter-dict helpers
# --------------------------------------------------------------------------
class Cfg(dict):
    """attribute dict used for the ``mtlora`` namespace (config.py:307-326, 477-557)."""
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):  # MODEL.DECODER_HEAD.get(task, 'hrnet') (swin_mtl.py:17)
        return dict.get(self, k, d)


def mtlora_config(tasks: Sequence[str], r_shared=64, r_task=4, scale=4.0, dropout=0.05,
                  shared_mode="matrix", n_stages=4, **over) -> Cfg:
    """What config.py:477-557 produces for the *_pertask yamls."""
    rl = {"shared": r_shared, **{t: r_task for t in tasks}}
    c = Cfg(ENABLED=True, QKV_ENABLED=True, PROJ_ENABLED=True, FC1_ENABLED=True, FC2_ENABLED=True,
            DOWNSAMPLER_ENABLED=False, INTERMEDIATE_SPECIALIZATION=False,
            TRAINABLE_SCALE_SHARED=False, TRAINABLE_SCALE_PER_TASK=False, SHARED_MODE=shared_mode,
            DROPOUT=[dropout] * n_stages, SHARED_SCALE=[scale] * n_stages,
            R_PER_TASK_LIST=[dict(rl) for _ in range(n_stages)],
            SCALE_PER_TASK_LIST=[{t: scale for t in tasks} for _ in range(n_stages)])
    c.update(over)
    return c


def det_tensor(name: str, shape: Sequence[int], scale: float = 0.02, offset: float = 0.0) -> Tensor:
    """Deterministic pseudo-random fill keyed by the parameter NAME (so the
    reference model, the oracle and the HIP model get identical weights without
    shipping a 100 MB state dict)."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    a = rng.standard_normal(tuple(shape)).astype(np.float32) * scale + offset
    return torch.from_numpy(a)


def det_fill_(named_tensors, skip_int=True) -> None:
    """In-place deterministic fill of every float tensor of an iterable of (name, tensor).
    LayerNorm/BatchNorm weights are centred on 1, running_var on 1, biases small."""
    with torch.no_grad():
        for n, p in named_tensors:
            if not torch.is_floating_point(p):
                continue
            if n.endswith("attn_mask"):
                continue
            last = n.split(".")[-1]
            if last == "running_var":
                p.copy_(det_tensor(n, p.shape, 0.05, 1.0).abs())
            elif last == "weight" and (n.split(".")[-2].startswith("norm") or ".last_layer.1." in n):
                p.copy_(det_tensor(n, p.shape, 0.05, 1.0))
            elif last in ("bias", "running_mean"):
                p.copy_(det_tensor(n, p.shape, 0.02))
            elif "lora_shared_scale" in n or "lora_task_scale" in n:   # TRAINABLE_SCALE_*: a scale of the usual magnitude
                p.copy_(det_tensor(n, p.shape, 0.5, 3.0))
            elif "lora_" in n and ("_A" in n):
                p.copy_(det_tensor(n, p.shape, 0.05))
            elif "lora_" in n:
                p.copy_(det_tensor(n, p.shape, 0.02))
            elif "downsample_" in n or "last_layer" in n or "patch_embed.proj" in n:
                p.copy_(det_tensor(n, p.shape, 0.05))
            else:
                p.copy_(det_tensor(n, p.shape, 0.02))


# --------------------------------------------------------------------------
# functional backbone (a5, a6, a10, a11)  over a flat parameter dict that uses
# the reference's state-dict names
# --------------------------------------------------------------------------
def _lin(P, pre, x, x_tasks, tasks, stage, mt, train, rng):
    """one MTLoRALinear / CompatLinear call from dict entries ``pre + ...``."""
    if pre + ".linear.weight" not in P:                         # CompatLinear (:36-41)
        return F.linear(x, P[pre + ".weight"], P.get(pre + ".bias")), None
    W, b = P[pre + ".linear.weight"], P.get(pre + ".linear.bias")
    has_tasks = tasks is not None and (pre + ".lora_tasks_A." + tasks[0]) in P
    A_s, B_s = P.get(pre + ".lora_shared_A"), P.get(pre + ".lora_shared_B")
    p = mt.DROPOUT[stage] if train else 0.0
    keep = None
    if p > 0.0 and A_s is not None and hasattr(rng, "keep_mask"):
        # replayed randomness (tests): the mask some other implementation drew for THIS call, in the row order of x
        keep = rng.keep_mask(pre, x)
    elif p > 0.0 and not x.is_cuda:
        keep = torch.rand(x.shape, generator=rng, device="cpu") >= p
    ss = P[pre + ".lora_shared_scale"] if (pre + ".lora_shared_scale") in P else mt.SHARED_SCALE[stage]
    return mtlora_linear(
        x, W, b, A_s, B_s, ss,
        tasks=list(tasks) if has_tasks else None,
        A_t={t: P[pre + ".lora_tasks_A." + t] for t in tasks} if has_tasks else None,
        B_t={t: P[pre + ".lora_tasks_B." + t] for t in tasks} if has_tasks else None,
        scale_t=mt.SCALE_PER_TASK_LIST[stage] if has_tasks else None,
        x_tasks=x_tasks if has_tasks else None,
        shared_mode=mt.SHARED_MODE if has_tasks else "matrix",
        keep_mask=keep, p=p,
        lora_norm=(P.get(pre + ".lora_norm.weight"), P.get(pre + ".lora_norm.bias")),
        torch_dropout=x.is_cuda and keep is None,
    )


def _ln(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + ".weight"], P[pre + ".bias"])


def _drop_path(x, p, train, rng, tag=None):
    """timm DropPath (per-sample stochastic depth, scale_by_keep) -- timm==0.9.2, third-party;
    call sites swin_transformer_mtlora.py:290-291, 390, 392, 399-407.  ``rng`` may be a replay object (tests) whose
    ``droppath(tag, x)`` returns the per-sample factors mask / keep another implementation drew for the residual ``tag``."""
    if p == 0.0 or not train:
        return x
    if hasattr(rng, "droppath"):
        f = rng.droppath(tag, x)
        return x * f.to(x.dtype).to(x.device).view((x.shape[0],) + (1,) * (x.ndim - 1))
    keep = 1.0 - p
    m = (torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), generator=rng) < keep).to(x.dtype).to(x.device)
    return x * m / keep


def swin_block(P, pre, x, H, W, num_heads, ws, shift, tasks, stage, mt, drop_path=0.0,
               train=False, rng=None):
    """SwinTransformerBlock.forward (:326-408) incl. WindowAttention (:186-227) and Mlp (:68-81)."""
    B, L, C = x.shape
    if min(H, W) <= ws:                                           # :271-274
        shift, ws = 0, min(H, W)
    shortcut = x
    xn = _ln(P, pre + ".norm1", x).reshape(B, H, W, C)
    xw = roll_and_window_partition(xn, shift, ws).reshape(-1, ws * ws, C)
    # --- WindowAttention
    qkv, _ = _lin(P, pre + ".attn.qkv", xw, None, None, stage, mt, train, rng)
    bias = dense_relative_bias(P[pre + ".attn.relative_position_bias_table"], ws)
    mask = shifted_window_mask(H, W, ws, shift)
    if mask is not None:
        mask = mask.to(x.dtype).to(x.device)
    a = window_attention_core(qkv, bias, mask, num_heads)
    aw, aw_t = _lin(P, pre + ".attn.proj", a, None, tasks, stage, mt, train, rng)
    # --- merge
    x = window_merge_and_roll(aw.reshape(-1, ws, ws, C), shift, ws, H, W).reshape(B, L, C)
    xt = None
    if aw_t is not None:                                          # :378-390
        xt = {}
        for t in tasks:
            m = window_merge_and_roll(aw_t[t].reshape(-1, ws, ws, C), shift, ws, H, W).reshape(B, L, C)
            xt[t] = shortcut + _drop_path(m, drop_path, train, rng, (pre, "attn", t))
    x = shortcut + _drop_path(x, drop_path, train, rng, (pre, "attn", None))           # :392
    # --- Mlp (:68-81)
    h, h_t = _lin(P, pre + ".mlp.fc1", _ln(P, pre + ".norm2", x),
                  {t: _ln(P, pre + ".norm2", xt[t]) for t in tasks} if xt is not None else None,
                  tasks, stage, mt, train, rng)
    h = F.gelu(h)
    if h_t is not None:
        h_t = {t: F.gelu(v) for t, v in h_t.items()}
    y, y_t = _lin(P, pre + ".mlp.fc2", h, h_t, tasks, stage, mt, train, rng)
    if y_t is None:                                               # :398-408
        return x + _drop_path(y, drop_path, train, rng, (pre, "mlp", None)), None
    out_t = {}
    for t in tasks:
        d = _drop_path(y_t[t], drop_path, train, rng, (pre, "mlp", t))
        out_t[t] = d if xt is None else xt[t] + d
    return x + _drop_path(y, drop_path, train, rng, (pre, "mlp", None)), out_t


def patch_merging(P, pre, x, H, W, stage, mt, train=False, rng=None):
    """PatchMerging.forward (:451-472)."""
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = _ln(P, pre + ".norm", x.reshape(B, -1, 4 * C))
    y, _ = _lin(P, pre + ".reduction", x, None, None, stage, mt, train, rng)
    return y


def backbone_stages(P, x, cfg, train=False, rng=None, prefix=""):
    """SwinTransformerMTLoRA.forward_features(return_stages=True) (:734-758), BasicLayer (:543-551),
    PatchEmbed (:597-605).  ``cfg``: img_size, patch_size, embed_dim, depths, num_heads,
    window_size, drop_path_rate, tasks, mtlora.  Returns [(x, {task: x_t})] per stage."""
    ps = cfg["patch_size"]
    x = F.conv2d(x, P[prefix + "patch_embed.proj.weight"], P[prefix + "patch_embed.proj.bias"], stride=ps)
    x = x.flatten(2).transpose(1, 2)
    if prefix + "patch_embed.norm.weight" in P:
        x = _ln(P, prefix + "patch_embed.norm", x)
    depths, heads, tasks, mt = cfg["depths"], cfg["num_heads"], cfg["tasks"], cfg["mtlora"]
    ws = cfg["window_size"]
    dpr = [v.item() for v in torch.linspace(0, cfg.get("drop_path_rate", 0.0), sum(depths))]  # :687-689
    H = W = cfg["img_size"] // ps
    out = []
    bi = 0
    for i, depth in enumerate(depths):
        tl = None
        for j in range(depth):
            lora = (j == depth - 1) or mt.INTERMEDIATE_SPECIALIZATION
            x, tl = swin_block(P, f"{prefix}layers.{i}.blocks.{j}", x, H, W, heads[i], ws,
                               0 if j % 2 == 0 else ws // 2, tasks if lora else None, i, mt,
                               dpr[bi], train, rng)
            bi += 1
        if i < len(depths) - 1:
            x = patch_merging(P, f"{prefix}layers.{i}.downsample", x, H, W, i, mt, train, rng)
            if tl is not None:
                tl = {t: patch_merging(P, f"{prefix}layers.{i}.downsample", v, H, W, i, mt, train, rng)
                      for t, v in tl.items()}
            H, W = H // 2, W // 2
        if tl is None:
            tl = {t: x for t in tasks}
        out.append((x, tl))
    return out


# --------------------------------------------------------------------------
# a13 callers: per-task Downsampler + HighResolutionHead + upsample + losses
# (swin_mtl.py:60-135, 223-246; seg_hrnet.py:498-526; mtl_loss_schemes.py)
# --------------------------------------------------------------------------
def _bn(P, pre, x, train):
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"],
                        P[pre + ".bias"], training=train, momentum=0.1, eps=1e-5)


def mtl_heads(P, stages, cfg, train=False):
    """MultiTaskSwin.forward after the backbone (swin_mtl.py:226-246)."""
    tasks = cfg["tasks"]
    n = len(cfg["depths"])
    res = [cfg["img_size"] // cfg["patch_size"] // (2 ** ((i + 1) if i < n - 1 else i)) for i in range(n)]
    out = {}
    for t in tasks:
        maps = []
        for i, (_, tl) in enumerate(stages):
            v = tl[t]
            m = v.reshape(-1, res[i], res[i], v.shape[-1]).permute(0, 3, 1, 2)
            maps.append(F.conv2d(m, P[f"downsampler.{t}.downsample_{i}.weight"]))
        h, w = maps[0].shape[2:]
        cat = torch.cat([maps[0]] + [F.interpolate(m, (h, w), mode="bilinear") for m in maps[1:]], 1)
        pre = f"decoders.decoders.{t}.last_layer"
        y = F.conv2d(cat, P[pre + ".0.weight"], P[pre + ".0.bias"])
        y = _bn(P, pre + ".1", y, train)
        if cfg.get("head_relu", True):  # (False: the kink-free probe of tests/test_gpu_models.py)
            y = F.relu(y)
        y = F.conv2d(y, P[pre + ".3.weight"], P[pre + ".3.bias"])
        out[t] = F.interpolate(y, (cfg["img_size"], cfg["img_size"]), mode="bilinear")
    return out


LOSS_WEIGHTS = {"depth": 1.0, "semseg": 1.0, "human_parts": 2.0, "sal": 5.0, "edge": 50.0, "normals": 10.0}  # main.py:192-199


def task_kind(task: str) -> str:
    """``t0``..``t7`` are the builder-defined synthetic tasks of BASELINE configs[4]: 3-channel regression
    heads under NormalsLoss with loss weight 1 (the reference's get_loss knows only its six named tasks)."""
    return "normals" if (len(task) >= 2 and task[0] == "t" and task[1:].isdigit()) else task


def num_output(task: str) -> int:
    return NUM_OUTPUT[task_kind(task)]


def loss_weight(task: str) -> float:
    return 1.0 if task_kind(task) != task else LOSS_WEIGHTS[task]


def task_loss(task: str, out: Tensor, label: Tensor) -> Tensor:
    """mtl_loss_schemes.py:241-263 dispatch."""
    task = task_kind(task)
    if task in ("semseg", "human_parts"):                        # SoftMaxwithLoss :22-39
        return F.nll_loss(F.log_softmax(out, 1), label[:, 0].long(), ignore_index=255)
    if task == "normals":                                        # NormalsLoss(normalize, L1) :162-220
        mask = label != 255
        n_valid = mask.sum().item()
        on = out / (torch.norm(out, p=2, dim=1).unsqueeze(1) + 1e-12)
        loss = F.l1_loss(torch.masked_select(on, mask), torch.masked_select(label, mask), reduction="sum")
        return loss / max(n_valid, 1e-6)
    if task == "sal":                                            # BalancedCrossEntropyLoss :42-89
        labels = (label >= 0.5).float()
        npos, nneg = labels.sum(), (1.0 - labels).sum()
        w = nneg / (npos + nneg)
        gz = (out >= 0).float()
        lv = out * (labels - gz) - torch.log(1 + torch.exp(out - 2 * out * gz))
        fl = w * (-(labels * lv)).sum() + (1 - w) * (-((1.0 - labels) * lv)).sum()
        return fl / float(np.prod(label.size()))
    if task == "depth":                                          # DepthLoss :132-148
        mask = label != 255
        return F.l1_loss(torch.masked_select(out, mask), torch.masked_select(label, mask))
    raise NotImplementedError(task)


def multi_task_loss(outs: Mapping[str, Tensor], targets: Mapping[str, Tensor], tasks: Sequence[str]):
    """MultiTaskLoss.forward (mtl_loss_schemes.py:232-238) with main.py:192-204 weights."""
    per = {t: task_loss(t, outs[t], targets[t]) for t in tasks}
    total = torch.sum(torch.stack([loss_weight(t) * per[t] for t in tasks]))
    return total, per


def full_model(P, x, cfg, train=False, rng=None):
    return mtl_heads(P, backbone_stages(P, x, cfg, train, rng, prefix="backbone."), cfg, train)


def trainable_filter(name: str) -> bool:
    """mark_only_lora_as_trainable with every freeze flag False and bias='none'
    (lora.py:580-630, main.py:257-262), for names under ``backbone.``; everything
    outside the backbone stays trainable."""
    if not name.startswith("backbone."):
        return True
    return any(s in name for s in ("lora_", "patch_embed", "norm", "downsample.reduction",
                                   "relative_position_bias_table"))


# --------------------------------------------------------------------------
# parameter-dict construction (shapes per SURVEY §8b; names = reference state dict)
# --------------------------------------------------------------------------
def backbone_param_shapes(cfg, prefix="") -> Dict[str, Tuple[int, ...]]:
    E, depths, heads, ws = cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window_size"]
    tasks, mt, ps = cfg["tasks"], cfg["mtlora"], cfg["patch_size"]
    res = cfg["img_size"] // ps
    S: Dict[str, Tuple[int, ...]] = {}
    S[prefix + "patch_embed.proj.weight"] = (E, cfg.get("in_chans", 3), ps, ps)
    S[prefix + "patch_embed.proj.bias"] = (E,)
    S[prefix + "patch_embed.norm.weight"] = (E,)
    S[prefix + "patch_embed.norm.bias"] = (E,)

    def lin(pre, K, N, stage, with_tasks, bias=True):
        S[pre + ".linear.weight"] = (N, K)
        if bias:
            S[pre + ".linear.bias"] = (N,)
        r = mt.R_PER_TASK_LIST[stage]
        if r["shared"] > 0:
            if mt.SHARED_MODE in ("matrix", "matrixv2") or not with_tasks:
                S[pre + ".lora_shared_A"] = (r["shared"], K)
                S[pre + ".lora_shared_B"] = (N, r["shared"])
            else:
                S[pre + ".lora_norm.weight"] = (N,)
                S[pre + ".lora_norm.bias"] = (N,)
            if mt.TRAINABLE_SCALE_SHARED:                       # lora.py:205-209: a 1-element Parameter
                S[pre + ".lora_shared_scale"] = (1,)
            if with_tasks:
                for t in tasks:
                    S[pre + ".lora_tasks_A." + t] = (r[t], K)
                    S[pre + ".lora_tasks_B." + t] = (N, r[t])

    for i, depth in enumerate(depths):
        C = E * 2 ** i
        w = min(ws, res // 2 ** i)
        for j in range(depth):
            b = f"{prefix}layers.{i}.blocks.{j}"
            lora = (j == depth - 1) or mt.INTERMEDIATE_SPECIALIZATION
            for nrm in ("norm1", "norm2"):
                S[f"{b}.{nrm}.weight"] = (C,)
                S[f"{b}.{nrm}.bias"] = (C,)
            S[f"{b}.attn.relative_position_bias_table"] = ((2 * w - 1) ** 2, heads[i])
            lin(f"{b}.attn.qkv", C, 3 * C, i, False)
            lin(f"{b}.attn.proj", C, C, i, lora)
            lin(f"{b}.mlp.fc1", C, 4 * C, i, lora)
            lin(f"{b}.mlp.fc2", 4 * C, C, i, lora)
        if i < len(depths) - 1:
            d = f"{prefix}layers.{i}.downsample"
            S[d + ".norm.weight"] = (4 * C,)
            S[d + ".norm.bias"] = (4 * C,)
            if mt.DOWNSAMPLER_ENABLED:
                lin(d + ".reduction", 4 * C, 2 * C, i, False, bias=False)
            else:
                S[d + ".reduction.weight"] = (2 * C, 4 * C)
    return S


def head_param_shapes(cfg, num_outputs: Mapping[str, int], channels=(18, 36, 72, 144)):
    E, n = cfg["embed_dim"], len(cfg["depths"])
    dims = [E * 2 ** ((i + 1) if i < n - 1 else i) for i in range(n)]
    S: Dict[str, Tuple[int, ...]] = {}
    cin = sum(channels)
    for t in cfg["tasks"]:
        for i in range(n):
            S[f"downsampler.{t}.downsample_{i}.weight"] = (channels[i], dims[i], 1, 1)
        pre = f"decoders.decoders.{t}.last_layer"
        S[pre + ".0.weight"] = (cin * 4, cin, 1, 1)
        S[pre + ".0.bias"] = (cin * 4,)
        for k in ("weight", "bias", "running_mean", "running_var"):
            S[f"{pre}.1.{k}"] = (cin * 4,)
        S[pre + ".3.weight"] = (num_outputs[t], cin * 4, 1, 1)
        S[pre + ".3.bias"] = (num_outputs[t],)
    return S


def make_params(shapes: Mapping[str, Tuple[int, ...]], dtype=torch.float32) -> Dict[str, Tensor]:
    P = {n: torch.empty(s, dtype=torch.float32) for n, s in shapes.items()}
    det_fill_(P.items())
    return {n: v.to(dtype) for n, v in P.items()}


def swin_t_cfg(img_size=448, tasks=("semseg", "normals", "sal", "human_parts"), r_shared=64, r_task=4,
               embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), drop_path_rate=0.2, **mt_over):
    return dict(img_size=img_size, patch_size=4, in_chans=3, embed_dim=embed_dim, depths=list(depths),
                num_heads=list(num_heads), window_size=7, drop_path_rate=drop_path_rate, tasks=list(tasks),
                mtlora=mtlora_config(tasks, r_shared, r_task, n_stages=len(depths), **mt_over))


NUM_OUTPUT = {"semseg": 21, "normals": 3, "sal": 1, "human_parts": 7, "depth": 1, "edge": 1}  # data/mtl_ds.py:749-780


def synthetic_batch(B: int, S: int, tasks: Sequence[str], seed: int, dtype=torch.float32):
    """SURVEY §8d synthetic inputs (shapes as collate delivers them, data/mtl_ds.py:861)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, S, S, generator=g)
    tg = {}
    for t in tasks:
        k = task_kind(t)
        if k in ("semseg", "human_parts"):
            lab = torch.randint(0, NUM_OUTPUT[k], (B, 1, S, S), generator=g).float()
            lab[torch.rand(B, 1, S, S, generator=g) < 0.05] = 255.0
        elif k == "sal":
            lab = (torch.rand(B, 1, S, S, generator=g) < 0.3).float()
        elif k == "normals":
            lab = F.normalize(torch.randn(B, 3, S, S, generator=g), dim=1)
            ign = (torch.rand(B, 1, S, S, generator=g) < 0.05).expand(B, 3, S, S)
            lab = torch.where(ign, torch.full_like(lab, 255.0), lab)
        elif k == "depth":
            lab = torch.rand(B, 1, S, S, generator=g) * 10
        else:
            raise NotImplementedError(t)
        tg[t] = lab.to(dtype)
    return img.to(dtype), tg
