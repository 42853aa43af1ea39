"""Callers of the hot path: the multi-task wrapper, the decoder heads, the losses and the
train step, restated in plain PyTorch so that ``bench.py`` / ``smoke()`` can run the reference's train step
on a GPU box that has no copy of the reference.

Nothing here is accelerated: the north star says the Swin-MTL wrapper and the semseg / normals / sal /
human_parts heads "drop in unchanged", i.e. a user keeps using the reference's ``models/swin_mtl.py``
and ``mtl_loss_schemes.py`` on top of ``mtlora_amd.swin_transformer_mtlora``.  The module / parameter
names below equal the reference's (``downsampler.{task}.downsample_{i}``, ``decoders.decoders.{task}.
last_layer.{0,1,3}``) so one state dict serves both.

Reference: models/swin_mtl.py:60-246, models/seg_hrnet.py:498-526, mtl_loss_schemes.py:22-263,
main.py:192-204 (loss weights), main.py:329-354 + utils.py:348-375 (train step).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from .functional import BatchNormReluFn, UpsampleLossFn
from .lora import MTLoRALinear, mark_only_lora_as_trainable
from .swin_transformer_mtlora import SwinTransformerMTLoRA

NUM_OUTPUT = {"semseg": 21, "normals": 3, "sal": 1, "human_parts": 7, "depth": 1, "edge": 1}  # data/mtl_ds.py:749-780
LOSS_WEIGHTS = {"depth": 1.0, "semseg": 1.0, "human_parts": 2.0, "sal": 5.0, "edge": 50.0, "normals": 10.0}


def is_synthetic_task(t: str) -> bool:
    """``t0`` .. ``t7``: the builder-defined tasks of BASELINE configs[4] (the reference knows only its six named
    tasks, mtl_loss_schemes.py:241-263) -- 3-channel regression heads trained with NormalsLoss, loss weight 1."""
    return len(t) >= 2 and t[0] == "t" and t[1:].isdigit()


def task_kind(t: str) -> str:
    return "normals" if is_synthetic_task(t) else t


def num_output(t: str) -> int:
    return NUM_OUTPUT[task_kind(t)]


def loss_weight(t: str) -> float:
    return 1.0 if is_synthetic_task(t) else LOSS_WEIGHTS[t]


# The BASELINE.json configs as ONE table (bench.py --config, the model-level parity tests and the docs all read it).
# c4 has no yaml in the reference ("builder authors it"): Swin-B = embed 128, depths 2-2-18-2, heads 4-8-16-32
# (the Swin paper's B variant through the reference constructor, swin_transformer_mtlora.py:643-649), r = 128 for the
# shared and every task factor.  c5:<r> is the 8-synthetic-task rank sweep, r for the shared and every task factor.
SWIN_T = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24))
SWIN_B = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
PASCAL4 = ("semseg", "normals", "sal", "human_parts")
CONFIGS = {
    "c1": dict(SWIN_T, img_size=224, tasks=("semseg",), r_shared=4, r_task=4, batch=2,
               what="BASELINE configs[0]: Swin-T 224, 1 task (semseg), r=4"),
    "c2": dict(SWIN_T, img_size=448, tasks=PASCAL4, r_shared=64, r_task=4, batch=32,
               what="BASELINE configs[1]: Swin-T 448, 4 tasks (semseg,normals,sal,human_parts), r_shared=64 r_task=4 scale4"),
    "c4": dict(SWIN_B, img_size=448, tasks=PASCAL4, r_shared=128, r_task=128, batch=16,
               what="BASELINE configs[3]: Swin-B 448 (embed 128, depths 2-2-18-2, heads 4-8-16-32), 4 tasks, r=128 shared and per task"),
}
for _r in (4, 16, 64, 256):
    CONFIGS[f"c5:{_r}"] = dict(SWIN_T, img_size=448, tasks=tuple(f"t{i}" for i in range(8)), r_shared=_r, r_task=_r, batch=32,
                               what=f"BASELINE configs[4]: Swin-T 448, 8 synthetic tasks t0..t7 (3-channel NormalsLoss heads), "
                                    f"r={_r} shared and per task")


def config(name: str) -> dict:
    """a copy of the CONFIGS row ``name`` (``c5:<r>`` accepts any positive rank)."""
    if name not in CONFIGS and name.startswith("c5:") and name[3:].isdigit() and int(name[3:]) > 0:
        r = int(name[3:])
        return dict(CONFIGS["c5:4"], r_shared=r, r_task=r, what=CONFIGS["c5:4"]["what"].replace("r=4 ", f"r={r} "))
    if name not in CONFIGS:
        raise KeyError(f"unknown config {name!r}; one of {sorted(CONFIGS)}")
    return dict(CONFIGS[name])


def build_config_model(name: str, seed: int = 0, **over) -> "MultiTaskSwin":
    c = config(name)
    c.update(over)
    return build_model(img_size=c["img_size"], tasks=c["tasks"], embed_dim=c["embed_dim"], depths=c["depths"],
                       num_heads=c["num_heads"], r_shared=c["r_shared"], r_task=c["r_task"], seed=seed,
                       **{k: v for k, v in c.items() if k not in ("img_size", "tasks", "embed_dim", "depths", "num_heads",
                                                                    "r_shared", "r_task", "batch", "what")})


class AttrDict(dict):
    """stand-in for the yacs CfgNode the reference constructors read attributes from."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def mtlora_namespace(tasks: Sequence[str], r_shared=64, r_task=4, scale=4.0, dropout=0.05, shared_mode="matrix",
                     n_stages=4, **over) -> AttrDict:
    """the ``config.MODEL.MTLORA`` namespace after config.py:477-557 normalisation (``*_pertask`` yamls)."""
    ns = AttrDict(ENABLED=True, QKV_ENABLED=True, PROJ_ENABLED=True, FC1_ENABLED=True, FC2_ENABLED=True,
                  DOWNSAMPLER_ENABLED=False, INTERMEDIATE_SPECIALIZATION=False, TRAINABLE_SCALE_SHARED=False,
                  TRAINABLE_SCALE_PER_TASK=False, SHARED_MODE=shared_mode, DROPOUT=[dropout] * n_stages,
                  SHARED_SCALE=[scale] * n_stages,
                  R_PER_TASK_LIST=[{"shared": r_shared, **{t: r_task for t in tasks}} for _ in range(n_stages)],
                  SCALE_PER_TASK_LIST=[{t: scale for t in tasks} for _ in range(n_stages)])
    ns.update(over)
    return ns


# --------------------------------------------------------------------------------------------------
_TASK_STREAMS = os.environ.get("MTLORA_TASK_STREAMS", "1") != "0"
# side streams (per-task heads, factor gradients) also while a HIP graph is being captured: the fork / join events become graph
# dependencies and the replay runs the branches concurrently.  MTLORA_GRAPH_STREAMS=0 captures a single-stream graph.
_GRAPH_STREAMS = os.environ.get("MTLORA_GRAPH_STREAMS", "1") != "0"


def _streams_allowed() -> bool:
    return _GRAPH_STREAMS or not torch.cuda.is_current_stream_capturing()


class Downsampler(nn.Module):
    """tokens (B, L_i, C_i) of the four stages -> NCHW maps through four bias-free 1x1 convs (swin_mtl.py:88-135).
    A 1x1 convolution on a token tensor is a per-token linear map, so it is evaluated as ``F.linear`` on the
    (B, L, C) tokens (hipBLASLt GEMM) and the result is only VIEWED as NCHW (channels-last strides): same
    parameters (``downsample_{i}.weight`` of shape (c, C, 1, 1)), same values, no MIOpen find / NCHW copy."""

    def __init__(self, dims, channels, input_res, bias=False, enabled=True):
        super().__init__()
        self.dims, self.input_res, self.enabled = dims, input_res, enabled
        if enabled:
            for i, (d, c) in enumerate(zip(dims, channels)):
                setattr(self, f"downsample_{i}", nn.Conv2d(d, c, 1, bias=bias))

    def forward(self, feats):
        maps = []
        for i, (f, r, d) in enumerate(zip(feats, self.input_res, self.dims)):
            if self.enabled:
                conv = getattr(self, f"downsample_{i}")
                B_, L_, _ = f.shape
                f = Fn.linear_big_m(f.reshape(B_ * L_, d), conv.weight.view(conv.out_channels, d), conv.bias).view(B_, L_, -1)
            maps.append(f.view(-1, r, r, f.shape[-1]).permute(0, 3, 1, 2))
        return maps


class HighResolutionHead(nn.Module):
    """upsample the 3 coarse maps to the finest, concat, 1x1 conv -> BN -> ReLU -> 1x1 conv (seg_hrnet.py:498-526).
    Same parameters and values as the reference's ``last_layer`` Sequential; the two 1x1 convolutions and the
    BatchNorm are evaluated on the (B*H*W, C) pixel matrix (channels-last), i.e. as GEMMs."""

    def __init__(self, backbone_channels, num_outputs):
        super().__init__()
        c = sum(backbone_channels)
        self.last_layer = nn.Sequential(nn.Conv2d(c, 4 * c, 1), nn.BatchNorm2d(4 * c, momentum=0.1),
                                        nn.ReLU(inplace=False), nn.Conv2d(4 * c, num_outputs, 1))

    def forward(self, x, channels_last_out: bool = False):
        B, _, Hh, Ww = x[0].shape
        c0, bn, _, c3 = self.last_layer
        w2d = c0.weight.view(c0.out_channels, c0.in_channels)
        xc = [m.permute(0, 2, 3, 1) for m in x]  # the Downsampler's maps are channels-last views
        chans = [t.shape[3] for t in xc]
        if (x[0].is_cuda and all(t.is_contiguous() for t in xc) and all(c % 4 == 0 for c in chans[1:])
                and len({t.dtype for t in xc}) == 1 and xc[0].dtype in (torch.float32, torch.bfloat16)
                and all(Hh % t.shape[1] == 0 and Ww % t.shape[2] == 0 and Hh // t.shape[1] == Ww // t.shape[2] for t in xc)):
            # upsample kernels write straight into the channel slices of the concatenated pixel matrix (padded so that
            # every slice starts at a multiple of 4 channels and rows are 16-byte aligned); the 1x1 conv weight gets the
            # matching zero columns
            t = Fn.ConcatUpsampleFn.apply(*xc)
            offs, ld = Fn.ConcatUpsampleFn.layout(chans)
            col_idx = None
            if ld != w2d.shape[1]:  # the weight's columns are scattered to the padded channel positions INSIDE the linear Function
                key = (tuple(offs), tuple(chans), w2d.device)  # (forward one index_copy, backward one gather next to the weight gradient)
                if getattr(self, "_col_key", None) != key:
                    self._col_idx = torch.cat([torch.arange(o, o + c) for o, c in zip(offs, chans)]).to(w2d.device)
                    self._col_key = key
                col_idx = self._col_idx
        else:
            cat = torch.cat([x[0]] + [F.interpolate(m, (Hh, Ww), mode="bilinear") for m in x[1:]], 1)
            t = cat.permute(0, 2, 3, 1).reshape(B * Hh * Ww, cat.shape[1])
            col_idx = None
        h = Fn.linear_big_m(t, w2d, c0.bias, feeds_batchnorm=bn.training, col_index=col_idx)
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        if bn.training and h.is_cuda and h.dtype in (torch.float32, torch.bfloat16, torch.float16) and h.shape[1] % 8 == 0:
            # fused training-mode BatchNorm + ReLU on the (pixels, channels) matrix (csrc/glue.hip)
            h = BatchNormReluFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                      getattr(self, "relu", True))
        else:
            h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training or not bn.track_running_stats,
                             bn.momentum, bn.eps)
            if getattr(self, "relu", True):  # (False only in the kink-free parity probe of tests/test_gpu_models.py)
                h = F.relu(h)
        w3, b3, nc = c3.weight.view(c3.out_channels, c3.in_channels), c3.bias, c3.out_channels
        # zero rows up to a multiple of 8 classes (the library's GEMM kernels take 16-byte rows): appended inside the linear Function
        o = Fn.linear_big_m(h, w3, b3, pad_rows=(8 - nc % 8) if (h.is_cuda and nc % 8) else 0)
        o = (o if o.shape[1] == nc else o[:, :nc].contiguous()).view(B, Hh, Ww, nc)
        return o if channels_last_out else o.permute(0, 3, 1, 2)


class DecoderGroup(nn.Module):
    def __init__(self, tasks, num_outputs, channels, out_size):
        super().__init__()
        self.tasks, self.out_size = tasks, out_size
        self.decoders = nn.ModuleDict({t: HighResolutionHead(channels, num_outputs[t]) for t in tasks})

    def forward(self, x, upsample: bool = True, tasks=None):
        """upsample=False: the low-resolution (B, h, w, C) predictions, for ``MultiTaskLoss.forward_low`` (the final
        bilinear upsample is fused into the loss kernels and never materialised)."""
        tasks = self.tasks if tasks is None else tasks
        if not upsample:
            return {t: self.decoders[t](x[t], channels_last_out=True) for t in tasks}
        return {t: F.interpolate(self.decoders[t](x[t]), self.out_size, mode="bilinear") for t in tasks}


class MultiTaskSwin(nn.Module):
    """backbone + per-task Downsampler + DecoderGroup (swin_mtl.py:138-246, MTLoRA-enabled, per-task downsampler,
    ``DECODER_HEAD: hrnet`` path -- the one every shipped config uses)."""

    def __init__(self, backbone: SwinTransformerMTLoRA, tasks: Sequence[str], num_outputs: Mapping[str, int] = NUM_OUTPUT,
                 decoder_channels=(18, 36, 72, 144)):
        super().__init__()
        self.backbone = backbone
        self.tasks = list(tasks)
        res = backbone.patch_embed.patches_resolution
        n = backbone.num_layers
        self.dims = [int(backbone.embed_dim * 2 ** ((i + 1) if i < n - 1 else i)) for i in range(n)]
        self.input_res = [res[0] // (2 ** ((i + 1) if i < n - 1 else i)) for i in range(n)]
        self.img_size = backbone.patch_embed.img_size
        self.downsampler = nn.ModuleDict({t: Downsampler(self.dims, decoder_channels, self.input_res) for t in self.tasks})
        self.decoders = DecoderGroup(self.tasks, num_outputs, decoder_channels, self.img_size)

    def task_streams(self, device):
        """one side stream per task (created once per device): the per-task Downsampler + head (+ fused loss) chains are
        independent, and half of their kernels are latency-bound (fused loss, upsample gathers, small GEMMs, reductions) --
        running the chains concurrently lets those hide under the other heads' bandwidth-bound GEMM / BatchNorm passes."""
        key = str(device)
        if getattr(self, "_streams_key", None) != key:
            self._streams = [torch.cuda.Stream(device=device) for _ in self.tasks]
            self._streams_key = key
        return self._streams

    def forward(self, x, upsample: bool = True, per_task_fn=None, concurrent: Optional[bool] = None):
        """per_task_fn(task, prediction) -> tensor: applied to each head's output inside that task's stream (train_step
        passes the fused per-task loss), its results are returned instead of the predictions.
        concurrent (default: MTLORA_TASK_STREAMS != 0, CUDA, grad mode): heads on per-task side streams; the outputs are
        joined back into the caller's stream before returning."""
        stages = self.backbone(x, return_stages=True)
        if concurrent is None:
            concurrent = (x.is_cuda and _TASK_STREAMS and len(self.tasks) > 1 and torch.is_grad_enabled()
                          and _streams_allowed())
        if not concurrent:
            feats = {t: self.downsampler[t]([tl[t] for _, tl in stages]) for t in self.tasks}
            out = self.decoders(feats, upsample=upsample)
            return out if per_task_fn is None else {t: per_task_fn(t, out[t]) for t in self.tasks}
        main = torch.cuda.current_stream(x.device)
        streams = self.task_streams(x.device)
        out = {}
        for t, st in zip(self.tasks, streams):
            ins = [tl[t] for _, tl in stages]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for f in ins:
                    f.record_stream(st)  # allocated on the caller's stream, read (and saved for backward) on the side stream
                y = self.decoders({t: self.downsampler[t](ins)}, upsample=upsample, tasks=[t])[t]
                if per_task_fn is not None:
                    y = per_task_fn(t, y)
                y.record_stream(main)
            out[t] = y
        for st in streams:
            main.wait_stream(st)
        return out


# --------------------------------------------------------------------------------------------------
# losses (mtl_loss_schemes.py), sync-free restatements: same value, no .item() / masked_select host round trip
# --------------------------------------------------------------------------------------------------
def task_loss(task: str, out: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    out = out.float()
    task = task_kind(task)
    if task in ("semseg", "human_parts"):          # SoftMaxwithLoss (:22-39)
        return F.nll_loss(F.log_softmax(out, 1), label[:, 0].long(), ignore_index=255)
    if task == "normals":                           # NormalsLoss(normalize=True, L1, size_average) (:162-220)
        mask = (label != 255).to(out.dtype)
        on = out / (torch.norm(out, p=2, dim=1, keepdim=True) + 1e-12)
        return ((on - label).abs() * mask).sum() / mask.sum().clamp_min(1e-6)
    if task == "sal":                               # BalancedCrossEntropyLoss(size_average) (:42-89)
        labels = (label >= 0.5).to(out.dtype)
        w = (1.0 - labels).sum() / labels.numel()
        gz = (out >= 0).to(out.dtype)
        lv = out * (labels - gz) - torch.log(1 + torch.exp(out - 2 * out * gz))
        return (w * (-(labels * lv)).sum() + (1 - w) * (-((1.0 - labels) * lv)).sum()) / labels.numel()
    if task == "depth":                             # DepthLoss l1 (:132-148)
        mask = (label != 255).to(out.dtype)
        return ((out - label).abs() * mask).sum() / mask.sum().clamp_min(1.0)
    raise NotImplementedError(task)


class MultiTaskLoss(nn.Module):
    """sum_t w_t * loss_t with the fixed weights of main.py:192-199 (mtl_loss_schemes.py:223-238)."""

    def __init__(self, tasks: Sequence[str], loss_weights: Optional[Mapping[str, float]] = None):
        super().__init__()
        self.tasks = list(tasks)
        self.loss_weights = dict(loss_weights or {t: loss_weight(t) for t in tasks})

    def forward(self, pred, gt):
        per = {t: task_loss(t, pred[t], gt[t]) for t in self.tasks}
        total = torch.stack([self.loss_weights[t] * per[t] for t in self.tasks]).sum()
        per["total"] = total
        return total, per

    FUSED_KIND = {"semseg": "softmax", "human_parts": "softmax", "normals": "normals", "sal": "balanced_bce"}

    def task_low(self, t, lo, lab):
        """loss of task t from its LOW-resolution (B, h, w, C) prediction (fused upsample + loss + backward where it applies)"""
        h, w = lo.shape[1:3]
        H, W = lab.shape[-2:]
        kind = self.FUSED_KIND.get(task_kind(t))
        if kind is not None and lo.is_cuda and H % h == 0 and W % w == 0 and H // h == W // w:
            return UpsampleLossFn.apply(kind, lo, lab, H // h)
        return task_loss(t, F.interpolate(lo.permute(0, 3, 1, 2), (H, W), mode="bilinear"), lab)

    def combine(self, per):
        per = dict(per)
        total = torch.stack([self.loss_weights[t] * per[t] for t in self.tasks]).sum()
        per["total"] = total
        return total, per

    def forward_low(self, low, gt):
        """same value and gradients as ``forward(upsampled prediction, gt)`` from the LOW-resolution (B, h, w, C)
        predictions of ``model(x, upsample=False)``: bilinear upsample + loss + backward fused (csrc/loss.hip).
        Tasks without a fused kernel (depth, edge) and non-integer scales take the plain path."""
        return self.combine({t: self.task_low(t, low[t], gt[t]) for t in self.tasks})


# --------------------------------------------------------------------------------------------------
def build_model(img_size=448, tasks=("semseg", "normals", "sal", "human_parts"), embed_dim=96, depths=(2, 2, 6, 2),
                num_heads=(3, 6, 12, 24), r_shared=64, r_task=4, drop_path_rate=0.2, lora_b_std=0.02, seed=0,
                freeze=True, **mt_over) -> MultiTaskSwin:
    """random-init model of the BASELINE configs: trunc_normal(.02) weights (reference :717-724),
    lora_*_B ~ N(0, lora_b_std) so that the LoRA paths are numerically live, and the reference's
    trainable set (mark_only_lora_as_trainable with every freeze flag False, main.py:257-262)."""
    torch.manual_seed(seed)
    mt = mtlora_namespace(tasks, r_shared, r_task, n_stages=len(depths), **mt_over)
    bb = SwinTransformerMTLoRA(img_size=img_size, patch_size=4, in_chans=3, num_classes=0, embed_dim=embed_dim,
                               depths=list(depths), num_heads=list(num_heads), window_size=7, mlp_ratio=4.0,
                               qkv_bias=True, drop_rate=0.0, drop_path_rate=drop_path_rate, ape=False, patch_norm=True,
                               tasks=list(tasks), mtlora=mt)
    model = MultiTaskSwin(bb, tasks, {t: num_output(t) for t in tasks})
    if lora_b_std > 0:
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "lora_" in n and "_B" in n:
                    p.normal_(0.0, lora_b_std)
    if freeze:
        mark_only_lora_as_trainable(model.backbone, bias="none")
    return model


def build_optimizer(model: nn.Module, lr=5e-4, weight_decay=0.05, fused: Optional[bool] = None,
                    capturable: bool = False) -> torch.optim.Optimizer:
    """AdamW(betas .9/.999, eps 1e-8, wd .05) with the no-decay set of optimizer.py:71-85 (1-D tensors, biases,
    relative_position_bias_table); only trainable tensors are handed over (frozen ones never get a grad in the
    reference either)."""
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim == 1 or n.endswith(".bias") or "relative_position_bias_table" in n or "absolute_pos_embed" in n:
            no_decay.append(p)
        else:
            decay.append(p)
    groups = [{"params": decay}, {"params": no_decay, "weight_decay": 0.0}]
    if fused is None:
        fused = all(p.is_cuda for p in decay + no_decay)
    return torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay, fused=fused,
                             capturable=bool(capturable and fused))


def synthetic_batch(B: int, S: int, tasks: Sequence[str], seed: int, device="cpu"):
    """SURVEY 8d synthetic inputs: image ~ N(0,1); semseg/human_parts class ids with 5 % 255; sal Bernoulli(.3);
    normals unit vectors with 5 % pixels 255."""
    g = torch.Generator(device="cpu").manual_seed(seed)
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
        tg[t] = lab.to(device)
    return img.to(device), tg


_FACTOR_STREAM = os.environ.get("MTLORA_FACTOR_STREAM", "1") != "0"
_factor_streams: Dict[str, "torch.cuda.Stream"] = {}


def _factor_side_stream(device):
    """the stream the MTLoRALinear factor gradients (k_tn reductions) run on during ``train_step``'s backward
    (functional.set_factor_stream); MTLORA_FACTOR_STREAM=0 keeps the backward on one stream."""
    if not _FACTOR_STREAM or not _streams_allowed():
        return None
    key = str(device)
    if key not in _factor_streams:
        _factor_streams[key] = torch.cuda.Stream(device=device)
    return _factor_streams[key]


_PACK_ONCE = os.environ.get("MTLORA_PACK_ONCE", "1") != "0"


class _NoPacker:
    def refresh(self):
        return 0


def _factor_packer(model):
    """the model's lora.FactorPacker (created on first use; MTLORA_PACK_ONCE=0: every forward packs its own factors)"""
    pk = getattr(model, "_mtlora_factor_packer", None)
    if pk is None:
        from .lora import FactorPacker
        pk = FactorPacker(model) if _PACK_ONCE else _NoPacker()
        object.__setattr__(model, "_mtlora_factor_packer", pk)
    return pk


def train_step(model, criterion, optimizer, images, targets, clip_grad: float = 5.0, reducer=None,
               amp_dtype: Optional[torch.dtype] = torch.bfloat16, fused_loss: bool = True):
    """one reference train step (main.py:329-354): autocast fwd + weighted multi-task loss, backward,
    [gradient all-reduce], clip_grad_norm_(5.0), AdamW, zero_grad.  bf16 autocast needs no GradScaler
    (the reference's scaler exists for its fp16 default)."""
    def fwd():
        if fused_loss:  # final upsample + losses (+ their backward) as one kernel per task, inside the task's stream
            if isinstance(model, MultiTaskSwin):
                return criterion.combine(model(images, upsample=False,
                                               per_task_fn=lambda t, lo: criterion.task_low(t, lo, targets[t])))
            return criterion.forward_low(model(images, upsample=False), targets)
        return criterion(model(images), targets)

    if images.is_cuda:
        Fn.droppath_begin_step(images.device)  # DropPath factors of the whole step from one draw (functional._DropPathPool)
        _factor_packer(model).refresh()        # the low-rank factors of every MTLoRALinear packed by ONE launch (lora.FactorPacker)
    try:
        if amp_dtype is not None:
            with torch.autocast("cuda", dtype=amp_dtype):
                loss, per = fwd()
        else:
            loss, per = fwd()
    finally:
        Fn.droppath_end_step()
    side = _factor_side_stream(images.device) if images.is_cuda else None
    if reducer is not None:
        extra = list(model._streams) if (isinstance(model, MultiTaskSwin) and getattr(model, "_streams", None)) else []
        reducer.extra_streams = extra + ([side] if side is not None else [])  # gradients produced off the main stream
        reducer.prepare()
    Fn.set_factor_stream(side)
    try:
        loss.backward()
    finally:
        Fn.set_factor_stream(None)
    if side is not None:
        torch.cuda.current_stream(images.device).wait_stream(side)  # dA / dB of every MTLoRALinear are complete from here on
        Fn.factor_stream_joined()
    if reducer is not None:
        reducer.finish()
    params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    norm = torch.nn.utils.clip_grad_norm_(params, clip_grad, foreach=True) if clip_grad else None
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return loss.detach(), norm


class GraphedTrainStep:
    """The SAME train step as ``train_step``, captured once as HIP graph(s) and replayed: the step issues ~1500 kernel
    launches plus the Python / autograd / ctypes work behind them (~50 ms of host time against ~55 ms of GPU time at
    C2), so eagerly it is nearly launch-bound -- and 8 ranks share one host.  A replay costs the host microseconds.

    * one graph = autocast forward + fused losses + backward + clip + AdamW + zero_grad (single rank), or
      backward graph (bucket packs captured) -> RCCL all-reduce of the flat buckets, EAGER, outside any graph ->
      optimizer graph (unpack + clip + AdamW) with a reducer (the collectives are deliberately not captured);
    * dropout: the MTLoRALinear seeds are baked into the graph, so a device int64 (``Fn.set_seed_offset``) is bumped
      by a captured add at the start of every replay and added to the seeds by the kernels at run time; DropPath
      uses torch's graph-safe generator;
    * data: ``images`` / ``targets`` are static buffers -- ``copy_`` the next batch into them before calling;
    * optimizer: must be ``capturable`` (``build_optimizer(..., capturable=True)``).
    Falls back to the eager step (``self.graphed = False``, reason in ``self.why``) if capture fails.

    Not bench.py's default (``--graph``): at C2 the eager step with its side streams is faster (34.8 ms vs 36.5 ms for the
    single-stream replay; the host is not the limit there), at C1 / small batches the replay is 2x the eager step.
    History: on the ROCm 7.0 / PyTorch 2.10 stack of this image a captured SMALL ``hipMemsetAsync`` stops taking effect from
    the second replay on (4 KB: stale data shows through; 4 MB is fine).  This library never issued memsets
    (common.h:mtl_zero_async) but ATen's multi-block reductions do; the nine per step (tools/find_memsets.py) now go through
    csrc/reduce.hip, the step issues none, and ``_replays_reproduce`` passes on every config -- it stays in place as the guard."""

    SEED_STEP = 0x1E3779B97F4A7C15  # odd: a full-period walk of the 64-bit seed offset

    def __init__(self, model, criterion, optimizer, images, targets, clip_grad: float = 5.0, reducer=None,
                 amp_dtype: Optional[torch.dtype] = torch.bfloat16, fused_loss: bool = True, warmup: int = 3):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.images, self.targets = images, targets
        self.clip_grad, self.reducer, self.amp_dtype, self.fused_loss = clip_grad, reducer, amp_dtype, fused_loss
        self.graphed, self.why = False, ""
        self.seed = torch.zeros(1, dtype=torch.int64, device=images.device)
        Fn.set_seed_offset(self.seed)
        self.g_bwd = self.g_opt = None
        self.loss = None
        try:
            self._capture(warmup)
            self.graphed = True
            ok, why = self._replays_reproduce()
            if not ok:
                raise RuntimeError(why)
        except Exception as e:  # noqa: BLE001 -- any capture / replay problem means: run eagerly, loudly
            self.graphed = False
            self.why = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
            self.g_bwd = self.g_opt = None
            optimizer.zero_grad(set_to_none=True)
            Fn.set_seed_offset(None)  # the eager fallback draws its dropout seeds on the host again

    def _replays_reproduce(self):
        """Replay the captured step three times FROM THE SAME STATE (parameters, optimizer state, RNG, dropout seed offset
        restored in between) and require bit-identical loss and parameters.  On this ROCm stack a captured small memset
        stops taking effect from the second replay on, so a whole-step graph (ATen's reductions issue such memsets) computes
        from stale memory after the first replay -- exactly what this catches.  The state is restored afterwards."""
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        bufs = [b for b in self.model.buffers() if b.is_floating_point()]

        def snap():
            st = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state.get(p, {}).items()}
                  for p in params}
            return ([p.detach().clone() for p in params], [b.clone() for b in bufs], st, torch.cuda.get_rng_state(),
                    self.seed.clone())

        def restore(s):
            with torch.no_grad():
                torch._foreach_copy_([p.data for p in params], s[0])
                if bufs:
                    torch._foreach_copy_(bufs, s[1])
                for p in params:
                    for k, v in s[2][id(p)].items():
                        if torch.is_tensor(v):
                            self.optimizer.state[p][k].copy_(v)
                self.seed.copy_(s[4])
            torch.cuda.set_rng_state(s[3])

        torch.cuda.synchronize()
        s0 = snap()
        seen = []
        for _ in range(3):
            restore(s0)
            self.g_bwd.replay()
            if self.g_opt is not None:
                self.reducer.all_reduce_packed()
                self.g_opt.replay()
            torch.cuda.synchronize()
            seen.append((self.loss.clone(), torch.stack([p.detach().double().sum() for p in params]).sum()))
        restore(s0)
        for l, c in seen[1:]:
            if not (torch.equal(l, seen[0][0]) and torch.equal(c, seen[0][1]) and torch.isfinite(l).all()):
                return False, ("graph replays from identical state differ (loss %r vs %r): captured small memsets do not replay "
                               "on this ROCm stack; running eagerly" % (seen[0][0].item(), l.item()))
        return True, ""

    # -- pieces of the step (shared by the eager fallback and the capture)
    def _forward_backward(self):
        self.seed.add_(self.SEED_STEP)
        if self.images.is_cuda:
            _factor_packer(self.model).refresh()  # (inside a capture: one graph node; the table and the masters keep their addresses)
        ctx = (torch.autocast("cuda", dtype=self.amp_dtype, cache_enabled=False) if self.amp_dtype is not None
               else contextlib.nullcontext())
        with ctx:
            if self.fused_loss and isinstance(self.model, MultiTaskSwin):  # as train_step: each loss inside its task's stream
                loss, _ = self.criterion.combine(self.model(
                    self.images, upsample=False, per_task_fn=lambda t, lo: self.criterion.task_low(t, lo, self.targets[t])))
            elif self.fused_loss:
                loss, _ = self.criterion.forward_low(self.model(self.images, upsample=False), self.targets)
            else:
                loss, _ = self.criterion(self.model(self.images), self.targets)
        side = _factor_side_stream(self.images.device) if self.images.is_cuda else None
        if self.reducer is not None:
            extra = list(self.model._streams) if (isinstance(self.model, MultiTaskSwin) and getattr(self.model, "_streams", None)) else []
            self.reducer.extra_streams = extra + ([side] if side is not None else [])
            self.reducer.prepare(defer=True)
        Fn.set_factor_stream(side)
        try:
            loss.backward()
        finally:
            Fn.set_factor_stream(None)
        if side is not None:  # joins the side stream back (inside a capture: closes the graph's second branch)
            torch.cuda.current_stream(self.images.device).wait_stream(side)
            Fn.factor_stream_joined()
        if self.reducer is not None:
            self.reducer.flush_packs()
        return loss.detach()

    def _optimize(self):
        if self.reducer is not None:
            self.reducer.unpack()
        params = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_(params, self.clip_grad, foreach=True)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)

    def _eager(self):
        loss = self._forward_backward()
        if self.reducer is not None:
            self.reducer.all_reduce_packed()
        self._optimize()
        return loss

    def _capture(self, warmup: int):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # allocator / autotune / lazy-init warm-up off the capture stream
            for _ in range(max(2, warmup)):  # (two: the FactorPacker builds its table -- a host-to-device copy -- at the SECOND step)
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.optimizer.zero_grad(set_to_none=True)
        self.g_bwd = torch.cuda.CUDAGraph()
        if self.reducer is None:
            with torch.cuda.graph(self.g_bwd):
                self.loss = self._forward_backward()
                self._optimize()
        else:
            with torch.cuda.graph(self.g_bwd):
                self.loss = self._forward_backward()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_bwd.pool()):
                self._optimize()
        torch.cuda.synchronize()

    def __call__(self):
        if not self.graphed:
            return self._eager()
        self.g_bwd.replay()
        if self.g_opt is not None:
            self.reducer.all_reduce_packed()
            self.g_opt.replay()
        return self.loss
