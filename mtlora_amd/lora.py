"""MI355X drop-in for ``models.lora`` (reference models/lora.py): ``LoRALayer``, ``MTLoRALinear``,
``mark_only_lora_as_trainable`` and ``map_old_state_dict_weights`` with the same constructor
arguments, attribute / parameter names and tuple-returning ``forward`` -- but forward and backward
run as fused gfx950 kernels (csrc/linear.hip) through the C ABI instead of 6+4T ATen launches.

Only the classes the reference actually instantiates are provided (``LoRALinear``, ``MTLoRAQKV`` and
``LoRAQKVLinear`` are dead code there).
"""
from __future__ import annotations

import math
import os
from typing import Any, Dict, Mapping, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import functional as Fn

_MODES = ("matrix", "matrixv2", "add", "addition", "lora_only")


class LoRALayer(nn.Module):
    """r, lora_alpha, dropout-or-identity, merged flag (reference lora.py:62-84)."""

    def __init__(self, r: int, lora_alpha: float, lora_dropout: float):
        super().__init__()
        assert r >= 0
        self.r = r
        self.lora_alpha = lora_alpha
        # kept for API parity; the HIP kernels apply the dropout themselves, reading `.p`
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else (lambda x: x)
        self.merged = False

    @property
    def dropout_p(self) -> float:
        d = self.lora_dropout
        return float(d.p) if isinstance(d, nn.Dropout) else 0.0


# Rank-aware association.  y = x W^T + s (x_in A^T) B^T costs 2 M r (K + N) flops and moves M x r intermediates
# (P forward, Q backward, both again for the factor gradients); for r > min(K, N) -- BASELINE configs[4] at r = 256: K = 96 / 192 in
# stages 0 / 1, up to nine outputs per layer, i.e. 2304 intermediate columns per token -- the update is cheaper as the K x N matrix it
# is.  The library's fused kernels take any pair (A', B') with y += (x_in A'^T) B'^T, so the update is handed over as
#     K <= N:  A' = I_K,        B' = s B A  (N x K)      [P' = D(x_in), Q' = dY Delta W]
#     K >  N:  A' = s B A,      B' = I_N                 [P' = the update itself, Q' = dY]
# i.e. "rank" min(K, N) instead of r, with exactly the reference's semantics (lora.py:253-284: dropout in front of A only, x_tasks
# layers see their own input).  Delta W is formed per call in fp32 by autograd-tracked matmuls, so dA = s B^T Z and dB = s Z A^T
# (Z = dB' or dA', the N x K gradient the library returns) come out of autograd; the identity side gets no gradient (and the
# library skips that reduction).  MTLORA_RANK_AWARE=0 keeps the low-rank form everywhere (A/B runs).
RANK_AWARE = os.environ.get("MTLORA_RANK_AWARE", "1") != "0"
_eyes: Dict[Any, torch.Tensor] = {}


def _delta_factors(As, Bs, scales, K: int, N: int):
    """effective (A', B') of updates s_i B_i A_i, each N x K: one batched matmul when the ranks agree"""
    dev = As[0].device
    key = (min(K, N), dev)
    eye = _eyes.get(key)
    if eye is None:
        eye = _eyes[key] = torch.eye(min(K, N), dtype=torch.float32, device=dev)
    with torch.autocast(dev.type, enabled=False):
        if len(As) > 1 and len({a.shape[0] for a in As}) == 1:
            sc = torch.tensor(scales, dtype=torch.float32, device=dev).view(-1, 1, 1) if len(set(scales)) > 1 else scales[0]
            dW = list((torch.bmm(torch.stack([b.float() for b in Bs]), torch.stack([a.float() for a in As])) * sc).unbind(0))
        else:
            dW = [(b.float() @ a.float()) * s for a, b, s in zip(As, Bs, scales)]
    if K <= N:
        return [eye] * len(As), dW
    return dW, [eye] * len(As)


class MTLoRALinear(LoRALayer):
    """Frozen ``linear`` + one shared and T per-task low-rank updates -> (y_shared, {task: y_task} | None).

    Same signature as reference lora.py:161-176; ``forward`` semantics of lora.py:253-284 (formulas in
    include/mtlora_hip.h)."""

    def __init__(self, in_features: int, out_features: int, r: Union[int, Mapping[str, int]] = 0,
                 lora_shared_scale: float = 1.0, lora_task_scale: Union[float, Mapping[str, float]] = 1.0,
                 lora_dropout: float = 0.0, tasks=None, trainable_scale_shared=False,
                 trainable_scale_per_task=False, shared_mode: str = "matrix", **kwargs):
        assert shared_mode in _MODES
        if shared_mode == "add":
            shared_mode = "addition"
        if shared_mode == "lora_only":
            tasks = None
        has_tasks = tasks is not None
        if not has_tasks and shared_mode != "matrix":
            shared_mode = "matrix"
        if isinstance(r, int):
            r = {"shared": r}
        super().__init__(r=r["shared"], lora_alpha=lora_shared_scale, lora_dropout=lora_dropout)
        self.linear = nn.Linear(in_features, out_features, **kwargs)
        self.tasks = tasks
        self.shared_mode = shared_mode
        self._ranks = dict(r)
        if r["shared"] > 0:
            w = self.linear.weight
            if has_tasks:
                self.lora_tasks_A = nn.ParameterDict({t: nn.Parameter(w.new_zeros((r[t], in_features))) for t in tasks})
                self.lora_tasks_B = nn.ParameterDict({t: nn.Parameter(w.new_zeros((out_features, r[t]))) for t in tasks})
                per_task = (lambda t: lora_task_scale[t]) if isinstance(lora_task_scale, Mapping) else (lambda t: lora_task_scale)
                if trainable_scale_per_task:
                    self.lora_task_scale = nn.ParameterDict(
                        {t: nn.Parameter(torch.FloatTensor([float(per_task(t))])) for t in tasks})
                else:
                    self.lora_task_scale = {t: lora_task_scale[t] for t in tasks}
            if shared_mode == "addition":
                assert has_tasks
                self.lora_norm = nn.LayerNorm(out_features)
            else:
                self.lora_shared_A = nn.Parameter(w.new_zeros((r["shared"], in_features)))
                self.lora_shared_B = nn.Parameter(w.new_zeros((out_features, r["shared"])))
            if trainable_scale_shared:
                self.lora_shared_scale = nn.Parameter(torch.FloatTensor([lora_shared_scale]))
            else:
                self.lora_shared_scale = lora_shared_scale
            self.reset_parameters()
        self._wcache: Dict[Any, Any] = {}
        # factors packed ahead of time by a FactorPacker (one launch per optimizer step for the whole model instead of one k_pack per
        # layer and forward): the buffer, what it was packed for (call signature + parameter versions), and the last call's signature
        self._packed: Optional[torch.Tensor] = None
        self._packed_sig = None
        self._call_sig = None

    def _factor_params(self):
        """the layer's low-rank factor Parameters (shared A, B, then the tasks' A..., B...); the tuple is built once -- Parameters are
        registered at construction and keep their identity through .to() / load_state_dict (both work in place)"""
        ps = self.__dict__.get("_fp_cache")
        if ps is None:
            ps = []
            if hasattr(self, "lora_shared_A"):
                ps += [self.lora_shared_A, self.lora_shared_B]
            if hasattr(self, "lora_tasks_A"):
                ps += [self.lora_tasks_A[t] for t in self.tasks] + [self.lora_tasks_B[t] for t in self.tasks]
            ps = tuple(ps)
            self.__dict__["_fp_cache"] = ps
        return ps

    def __setattr__(self, name, value):
        if name.startswith("lora_"):  # a factor (container) is being (re)assigned: the cached tuple of _factor_params is stale
            self.__dict__.pop("_fp_cache", None)
        super().__setattr__(name, value)

    def reset_parameters(self):
        """A ~ kaiming_uniform(a=sqrt 5), B = 0 (reference lora.py:236-247)."""
        if hasattr(self, "lora_shared_A"):
            nn.init.kaiming_uniform_(self.lora_shared_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_shared_B)
        if hasattr(self, "lora_tasks_A"):
            for t in self.tasks:
                nn.init.kaiming_uniform_(self.lora_tasks_A[t], a=math.sqrt(5))
                nn.init.zeros_(self.lora_tasks_B[t])

    # -- inference-time merging.  The reference's ``merge`` is a stub that raises (lora.py:249-251); SURVEY 8f row 4 asks for
    # the real thing: W' = W + s_s B_s A_s.  One weight can only stand for the layer when every output sees the shared update:
    # layers without tasks (36 of the 48 calls of a Swin-T forward) and ``matrixv2`` layers.  With ``shared_mode='matrix'`` and
    # tasks, y_t = x W^T + s_t (x_t A_t^T) B_t^T must NOT contain the shared update, so such a layer stays unmerged.
    def _mergeable(self) -> bool:
        return (self.r > 0 and hasattr(self, "lora_shared_A")
                and (self.tasks is None or self.shared_mode == "matrixv2"))

    def _shared_delta(self) -> torch.Tensor:
        s = self.lora_shared_scale
        s = float(s.detach().item()) if isinstance(s, torch.Tensor) else float(s)
        return (self.lora_shared_B.detach().float() @ self.lora_shared_A.detach().float()) * s

    def merge(self) -> bool:
        """fold the shared low-rank update into ``linear.weight`` (eval only: the train-time dropout in front of A cannot be
        merged).  Returns True when the layer was merged; ``unmerge`` / ``train()`` undo it."""
        if self.merged or not self._mergeable():
            return False
        if self.training:
            raise RuntimeError("mtlora_amd: merge() folds the shared update into the frozen weight for inference; call .eval() "
                               "first (the train-time dropout in front of A cannot be merged)")
        with torch.no_grad():
            self.linear.weight.data.add_(self._shared_delta().to(self.linear.weight.dtype))
        self.merged = True
        self.invalidate_weight_cache()
        return True

    def unmerge(self) -> bool:
        if not self.merged:
            return False
        with torch.no_grad():
            self.linear.weight.data.sub_(self._shared_delta().to(self.linear.weight.dtype))
        self.merged = False
        self.invalidate_weight_cache()
        return True

    # -- frozen-weight copies in the compute dtype (W, W^T) and an fp32 bias, refreshed when W changes
    def _weights(self, dtype: torch.dtype):
        w, b = self.linear.weight, self.linear.bias
        key = (dtype, w.device)
        ver = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()))
        hit = self._wcache.get(key)
        if w.requires_grad or (b is not None and b.requires_grad):
            hit = None  # trained by an optimizer every step: never serve a cached copy
        if hit is not None and hit[1].is_inference() and not torch.is_inference_mode_enabled():
            hit = None  # copies made under torch.inference_mode() (an eval pass) cannot be saved for backward or refreshed in place
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                if hit is not None and hit[1].shape == w.shape and (b is None) == (hit[3] is None):
                    # refresh IN PLACE: a captured HIP graph (GraphedTrainStep) has the addresses of these copies baked in
                    wc, wt, bf = hit[1], hit[2], hit[3]
                    wc.copy_(w.detach())
                    wt.copy_(w.detach().t())
                    if bf is not None:
                        bf.copy_(b.detach())
                else:
                    wc = w.detach().to(dtype).contiguous()
                    wt = w.detach().t().to(dtype).contiguous()
                    bf = None if b is None else b.detach().float().contiguous()
            hit = (ver, wc, wt, bf)
            # one entry per (dtype, device), none evicted: an fp32 eval between bf16 graphed steps must not free the copies whose
            # addresses a captured HIP graph holds (52 MB per extra dtype at C2: irrelevant against 288 GB)
            self._wcache[key] = hit
        return hit[1], hit[2], hit[3]

    def invalidate_weight_cache(self) -> None:
        """mark the cached compute-dtype copies of ``linear.weight`` / ``linear.bias`` stale: the next forward re-fills the SAME
        buffers (their addresses may be baked into a captured HIP graph).  The cache keys on the tensors' version counters and
        storage, which in-place updates through ``.data`` (EMA, hand-written loaders, older optimizers) do NOT bump: call this
        after such an update.  ``load_state_dict``, ``merge`` / ``unmerge`` call it themselves; ``.to()`` / ``.cuda()`` drop the
        copies altogether.  The same holds for the low-rank factors packed ahead of time by a ``FactorPacker``: an edit of
        ``lora_*`` through ``.data`` needs this call too (it drops the "packed factors are current" mark; the next forward packs
        inside the call until the packer refreshes)."""
        self._wcache = {k: (None,) + tuple(v[1:]) for k, v in self._wcache.items()}
        self._packed_sig = None  # the packed low-rank factors too (freshness is judged by Parameter._version, which .data edits skip)

    def _apply(self, fn, *a, **k):  # .to() / .cuda(): the copies live on the old device / dtype
        self._wcache = {}
        self._packed_sig = None
        self.__dict__.pop("_fp_cache", None)
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        # a state dict always holds the UN-merged pretrained weight (see state_dict below): whatever is loaded replaces a merged
        # weight, so the flag goes back to "not merged" (loading into a merged module used to leave the flag set: the shared update
        # was then skipped, and the next train() subtracted a delta the loaded weight never contained)
        # -- unless the dict is PARTIAL (strict=False: LoRA factors only, checkpoint.load_state): then linear.weight still holds
        # W + s B A of the OLD factors, and clearing the flag would make the next forward add the shared update a second time while
        # train() / unmerge() no longer subtracted it: un-merge with the old factors first, then load.
        if self.merged:
            if (prefix + "linear.weight") in state_dict:
                self.merged = False
            else:
                self.unmerge()
        self.invalidate_weight_cache()
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """a merged layer saves the UN-merged pretrained weight next to its factors (saving W + s B A together with A and B would
        apply the shared update twice after a reload)."""
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        if self.merged:
            key = prefix + "linear.weight"
            if key in sd:
                w = sd[key]
                sd[key] = (w.detach() - self._shared_delta().to(w.dtype)) if not keep_vars else w - self._shared_delta().to(w.dtype)
        return sd

    def train(self, mode: bool = True):
        if mode and self.merged:  # a merged weight cannot be trained through (loralib convention: train() un-merges)
            self.unmerge()
        return super().train(mode)  # (the cached copies stay: W did not change, and a captured graph may hold their addresses)

    def _rank_aware(self, r: int, scale) -> bool:
        """apply an update of rank ``r`` as Delta W = s B A?  Only when that SHRINKS the intermediates (r > min(K, N)) and the scale
        is a constant (a trainable scale gets its gradient from <dB, B> / s of the low-rank form)."""
        return (RANK_AWARE and r > min(self.linear.in_features, self.linear.out_features) and not isinstance(scale, torch.Tensor)
                and not self.merged)

    def rank_aware_shared(self) -> bool:
        return self.r > 0 and hasattr(self, "lora_shared_A") and self._rank_aware(self.r, self.lora_shared_scale)

    def meta_t0(self, dtype: torch.dtype, device) -> "Fn.LinearMeta":
        """the LinearMeta of a call of a layer WITHOUT tasks and with a constant shared scale (``tasks is None``, shared_mode 'matrix',
        not merged) -- what ``forward`` builds for such a layer, for callers that issue the library call themselves (the one-call
        Swin block, swin_transformer_mtlora.SwinTransformerBlock._block_call).  Draws the call's dropout seed."""
        p = self.dropout_p if self.training else 0.0
        ss = float(self.lora_shared_scale)
        meta = Fn.LinearMeta(K=self.linear.in_features, N=self.linear.out_features, r_s=self.r, r_t=(), scale_s=ss, scale_t=(), mode=0,
                             has_x_tasks=False, dropout_p=p, seed=Fn.next_seed() if p > 0 else 0, dtype=dtype)
        sig = (dtype, meta.r_s, (), ss, (), 0, False, p, str(device))
        self._call_sig = sig
        if self._packed is not None and self._packed_sig == (sig, tuple(q._version for q in self._factor_params())):
            meta.packed = self._packed
        return meta

    def hid_call(self, dtype: torch.dtype, device):
        """(meta, weights, factors) of a call WITH x_tasks of a task-enabled layer in shared_mode 'matrix' -- what ``forward`` builds for
        such a layer, for the Mlp that issues fc1 / fc2 itself with the task hidden tensors left implicit (``Fn.MlpHidFn``); None when
        the layer does not qualify (no tasks, merged, trainable scales, rank-aware association, trained pretrained weight).  Draws the
        call's dropout seed."""
        if not (self.r > 0 and self.tasks and self.shared_mode == "matrix" and not self.merged and hasattr(self, "lora_tasks_A")
                and hasattr(self, "lora_shared_A")):
            return None
        tasks = list(self.tasks)
        if isinstance(self.lora_shared_scale, torch.Tensor) or any(isinstance(self.lora_task_scale[t], torch.Tensor) for t in tasks):
            return None
        if self.linear.weight.requires_grad or (self.linear.bias is not None and self.linear.bias.requires_grad):
            return None
        if self.rank_aware_shared() or any(self._rank_aware(self._ranks[t], self.lora_task_scale[t]) for t in tasks):
            return None
        p = self.dropout_p if self.training else 0.0
        ss, st = float(self.lora_shared_scale), tuple(float(self.lora_task_scale[t]) for t in tasks)
        meta = Fn.LinearMeta(K=self.linear.in_features, N=self.linear.out_features, r_s=self.r, r_t=tuple(self._ranks[t] for t in tasks),
                             scale_s=ss, scale_t=st, mode=0, has_x_tasks=True, dropout_p=p, seed=Fn.next_seed() if p > 0 else 0, dtype=dtype)
        sig = (dtype, meta.r_s, meta.r_t, meta.scale_s, meta.scale_t, meta.mode, True, p, str(device))  # (as forward: FactorPacker key)
        self._call_sig = sig
        if self._packed is not None and self._packed_sig == (sig, tuple(q._version for q in self._factor_params())):
            meta.packed = self._packed
        factors = (self.lora_shared_A, self.lora_shared_B, [self.lora_tasks_A[t] for t in tasks], [self.lora_tasks_B[t] for t in tasks])
        return meta, self._weights(dtype), factors

    def forward(self, x: torch.Tensor, x_tasks: Optional[Dict[str, torch.Tensor]] = None, gelu_gate=None, gelu_out: bool = False
                ) -> Tuple[torch.Tensor, Optional[Dict[str, torch.Tensor]]]:
        """gelu_gate = (h, {task: h_t} or None): x = gelu(h) and x_tasks[t] = gelu(h_t) came from a deferred-gradient GELU
        (``gelu_out`` of the producing layer, or ``Fn.GeluDeferredGradFn``: identity backward); this layer's backward then
        returns the gradients w.r.t. h.
        gelu_out=True: returns ``(y, y_tasks, gelu(y), {t: gelu(y_tasks[t])})``, the activations written by the same kernel;
        a gradient reaching them is taken as a gradient w.r.t. y (see ``gelu_gate``): only for that pairing."""
        dtype = Fn.compute_dtype(x)  # (device checks: MTLoRALinearFn.forward)
        wc, wt, bf = self._weights(dtype)
        has_lora = self.r > 0
        tasks = list(self.tasks) if (has_lora and self.tasks is not None) else []
        shared = has_lora and self.shared_mode in ("matrix", "matrixv2") and not self.merged
        p = self.dropout_p if (self.training and has_lora) else 0.0
        val = lambda s: float(s.detach().item()) if isinstance(s, torch.Tensor) else float(s)
        par = lambda s: s if isinstance(s, nn.Parameter) else None
        ss = self.lora_shared_scale if shared else 0.0
        st = [self.lora_task_scale[t] for t in tasks]
        # rank-aware association (round 6): an update of rank r > min(K, N) is applied as the (K x N)-sized Delta W = s B A it is
        # (``_delta_factors``): the M x r intermediates of the low-rank form become M x min(K, N)
        ra_s = shared and self.rank_aware_shared()
        ra_t = [bool(tasks) and self._rank_aware(self._ranks[t], st[0]) for t in tasks]
        A_s, B_s = (self.lora_shared_A, self.lora_shared_B) if shared else (None, None)
        A_t, B_t = [self.lora_tasks_A[t] for t in tasks], [self.lora_tasks_B[t] for t in tasks]
        r_s, r_t = (self.r if shared else 0), [self._ranks[t] for t in tasks]
        vs, vt = val(ss), [val(s) for s in st]
        if ra_s or any(ra_t):
            K_, N_ = self.linear.in_features, self.linear.out_features
            if ra_s:
                (A_s,), (B_s,) = _delta_factors([A_s], [B_s], [vs], K_, N_)
                r_s, vs = min(K_, N_), 1.0
            idx = [i for i, f in enumerate(ra_t) if f]
            if idx:
                Ae, Be = _delta_factors([A_t[i] for i in idx], [B_t[i] for i in idx], [vt[i] for i in idx], K_, N_)
                for j, i in enumerate(idx):
                    A_t[i], B_t[i], r_t[i], vt[i] = Ae[j], Be[j], min(K_, N_), 1.0
        meta = Fn.LinearMeta(
            K=self.linear.in_features, N=self.linear.out_features,
            r_s=r_s, r_t=tuple(r_t), scale_s=vs, scale_t=tuple(vt),
            mode=1 if (self.shared_mode == "matrixv2" and tasks and not self.merged) else 0,
            has_x_tasks=bool(tasks) and x_tasks is not None, dropout_p=p, seed=Fn.next_seed() if p > 0 else 0,
            dtype=dtype, weight_requires_grad=self.linear.weight.requires_grad or (
                self.linear.bias is not None and self.linear.bias.requires_grad),
            n_scale_t=len(tasks) if (tasks and isinstance(st[0], nn.Parameter)) else 0)
        if has_lora and (ra_s or any(ra_t)):
            self._call_sig = None  # (the effective factors are formed per call: nothing for a FactorPacker to pack ahead of time)
        elif has_lora:
            # (what the packed factors depend on besides the masters: the alpha-scaled copies carry the scales and the dropout scale)
            sig = (dtype, meta.r_s, meta.r_t, meta.scale_s, meta.scale_t, meta.mode, meta.has_x_tasks, p, str(x.device))
            self._call_sig = sig
            if self._packed is not None and self._packed_sig == (sig, tuple(q._version for q in self._factor_params())):
                meta.packed = self._packed
        if gelu_out:
            if self.shared_mode == "addition":
                raise RuntimeError("mtlora_amd: gelu_out is not available with shared_mode='addition'")
            meta.gelu_out = True
        gates = []
        if gelu_gate is not None:
            gates = [gelu_gate[0]] + ([gelu_gate[1][t] for t in tasks] if meta.has_x_tasks else [])
            meta.n_gate = len(gates)
        args = [meta, x, wc, wt, bf, self.linear.weight, self.linear.bias, A_s, B_s, None if ra_s else par(ss)]
        if meta.has_x_tasks:
            args += [x_tasks[t] for t in tasks]
        args += A_t + B_t
        if meta.n_scale_t:
            args += st
        args += gates
        outs = Fn.MTLoRALinearFn.apply(*args)
        y = outs[0]
        if gelu_out:
            nt = len(tasks)
            return (y, {t: outs[1 + i] for i, t in enumerate(tasks)} if tasks else None,
                    outs[1 + nt], {t: outs[2 + nt + i] for i, t in enumerate(tasks)} if tasks else None)
        if not has_lora:
            return y, None
        y_tasks = {t: outs[1 + i] for i, t in enumerate(tasks)} if tasks else None
        if self.shared_mode == "addition":  # lora.py:275-282: LayerNorm(sum_t y_t) added to the pretrained output
            tot = torch.stack(list(y_tasks.values()), 0).sum(0).float()
            y = y + nn.functional.layer_norm(tot, (tot.shape[-1],), self.lora_norm.weight.float(),
                                             self.lora_norm.bias.float(), self.lora_norm.eps).to(y.dtype)
        return y, y_tasks


class FactorPacker:
    """Packs the low-rank factors of every MTLoRALinear of ``model`` in ONE kernel launch (``mtlora_linear_pack_table``) instead of one
    ``k_pack`` launch per layer and forward call (48 per Swin-T step, 96 per Swin-B step).  The factors change when the optimizer
    steps; a trainer calls ``refresh()`` once per step before the forward (``mtl_harness.train_step`` does).  A layer uses its packed
    buffer only while it is provably current: same call signature (dtype, ranks, scales, dropout p, x_tasks or not) and unchanged
    ``_version`` of every factor Parameter -- anything else (first step, eval, a state-dict load, hand edits) falls back to packing
    inside the forward call.  Layers with trainable scales (1-element Parameters that move every step) are left out."""

    def __init__(self, model: nn.Module):
        self.layers = [m for m in model.modules() if isinstance(m, MTLoRALinear) and m.r > 0]
        self._table = None
        self._table_key = None
        # every table ever built and every packed buffer it points to, per key: a captured HIP graph (GraphedTrainStep) has the raw
        # device addresses of the table and of the layers' packed buffers baked into its nodes, so a change of call signature
        # (an eval or fp32 forward between graphed steps) must not free them -- same policy as MTLoRALinear._wcache
        self._tables = {}

    def _eligible(self, m: "MTLoRALinear") -> bool:
        if m._call_sig is None or m.merged:
            return False
        if isinstance(m.lora_shared_scale, torch.Tensor):
            return False
        if m.tasks is not None and isinstance(m.lora_task_scale, nn.ParameterDict):
            return False
        return all(q.dtype == torch.float32 and q.is_contiguous() and q.is_cuda for q in m._factor_params())

    def refresh(self) -> int:
        """(re)pack every eligible layer for the signature of its LAST forward call; returns the number of layers packed."""
        for m in self.layers:  # (an entry of a ParameterDict replaced behind the module's back is picked up here)
            m.__dict__.pop("_fp_cache", None)
        todo = [m for m in self.layers if self._eligible(m)]
        if not todo:
            return 0
        key = tuple((id(m), m._call_sig) + tuple(q.data_ptr() for q in m._factor_params()) for m in todo)
        if key != self._table_key and key in self._tables:
            self._table = self._tables[key]
            self._table_key = key
            for tb in self._table:  # the layers go back to the buffers this table writes
                for ent in tb.keep:
                    ent[6]._packed = ent[5]
        elif key != self._table_key:
            by_dtype = {}
            for m in todo:
                dtype, r_s, r_t, scale_s, scale_t, mode, has_xt, p, _dev = m._call_sig
                meta = Fn.LinearMeta(K=m.linear.in_features, N=m.linear.out_features, r_s=r_s, r_t=r_t, scale_s=scale_s, scale_t=scale_t,
                                     mode=mode, has_x_tasks=has_xt, dropout_p=p, seed=0, dtype=dtype)
                dev = m._factor_params()[0].device
                need = Fn.packed_bytes(meta)
                # one buffer per (layer, signature), never freed or re-used for another signature (see __init__)
                m._packed = torch.empty(need, dtype=torch.uint8, device=dev)
                shared = r_s > 0
                tasks = list(m.tasks) if (m.tasks is not None and len(r_t) > 0) else []
                ent = (meta, m.lora_shared_A if shared else None, m.lora_shared_B if shared else None,
                       [m.lora_tasks_A[t] for t in tasks], [m.lora_tasks_B[t] for t in tasks], m._packed, m)
                by_dtype.setdefault((dtype, str(dev)), []).append(ent)
            self._table = [Fn.PackTable(ents, ents[0][5].device, dt) for (dt, _), ents in by_dtype.items()]
            self._table_key = key
            self._tables[key] = self._table
        for tb in self._table:
            tb.pack()
        for m in todo:
            m._packed_sig = (m._call_sig, tuple(q._version for q in m._factor_params()))
        return len(todo)


def mark_only_lora_as_trainable(model: nn.Module, bias: str = "none", freeze_patch_embed: bool = False,
                                freeze_norm: bool = False, free_relative_bias: bool = False,
                                freeze_downsample_reduction=False) -> None:
    """Same filters, flags (incl. the inverted-sounding ``free_relative_bias``) and bias modes as reference
    lora.py:580-630: a parameter stays trainable iff its name contains ``lora_`` or one of the un-frozen
    families (patch_embed / norm / downsample.reduction / relative_position_bias_table)."""
    keep = [("lora_", True), ("patch_embed", not freeze_patch_embed), ("norm", not freeze_norm),
            ("downsample.reduction", not freeze_downsample_reduction),
            ("relative_position_bias_table", not free_relative_bias)]
    for name, p in model.named_parameters():
        if not any(on and key in name for key, on in keep):
            p.requires_grad = False
    if bias == "none":
        return
    if bias == "all":
        for name, p in model.named_parameters():
            if "bias" in name:
                p.requires_grad = True
    elif bias == "lora_only":
        for m in model.modules():
            if isinstance(m, LoRALayer) and getattr(m, "bias", None) is not None:
                m.bias.requires_grad = True
    else:
        raise NotImplementedError


def lora_filter(key: str, value: Any) -> bool:
    return "lora_" in key


def map_old_state_dict_weights(state_dict: Dict, mapping: Mapping, prefix: str, split_qkv: bool = False) -> Dict:
    """Rename vanilla-Swin checkpoint keys to the MTLoRA layout (``attn.qkv.weight`` -> ``attn.qkv.linear.weight``),
    optionally splitting fused qkv rows into q/k/v (reference lora.py:644-668)."""
    missing = []
    for old, new in mapping.items():
        src = prefix + old
        if src not in state_dict:
            missing.append(old)
            continue
        w = state_dict.pop(src)
        dst = prefix + new
        tail = ".".join(dst.split(".")[-4:])
        if split_qkv and tail in ("attn.qkv.linear.weight", "attn.qkv.linear.bias"):
            kind = tail.split(".")[-1]
            stem = ".".join(dst.split(".")[:-2])
            for nm, part in zip("qkv", torch.chunk(w, chunks=3)):
                state_dict[f"{stem}.{nm}.linear.{kind}"] = part
        else:
            state_dict[dst] = w
    if missing:
        print(f"WARNING: The following keys from the checkpoint were not mapped: {missing}")
    return state_dict
