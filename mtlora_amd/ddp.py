"""Data-parallel gradient exchange for the MTLoRA train step: one process per GPU, RCCL over xGMI.

The reference has NO gradient synchronisation at all (it initialises NCCL and calls one barrier,
main.py:566-568; the model is never wrapped, main.py:167) -- SURVEY 2.3.  This is therefore new
behaviour, specified in SURVEY 8e: replicas each take their own synthetic shard and the gradients of the
TRAINABLE parameters only (LoRA factors, norms, relative-position tables, patch embed, downsample
reductions, decoder heads: 8.34 M fp32 = 33.4 MB at C2; the 26 M frozen W never move) are averaged with an
all-reduce once per step.

Design for 8 fully-connected MI355X (7 xGMI links x ~153 GB/s per GPU): 33 MB is far too small to be
bandwidth-bound (~0.1-0.4 ms), so the cost is launch latency and exposure.  Gradients are packed into a few
large flat fp32 buckets in REVERSE parameter order (the order backward produces them: heads first, stage-0
LoRA last); a bucket's all-reduce is launched asynchronously from the autograd hook of its last-arriving
gradient, so everything but the final bucket overlaps the remaining backward kernels.  Parameters that get
no gradient in a step (the final stage's shared fc2 factors are never used) simply contribute
zeros and keep ``.grad is None`` -- no ``find_unused_parameters`` graph walk.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "handle", "seen", "arrived", "launched")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        self.pending = len(params)
        self.handle = None
        self.seen = [False] * len(params)      # has a gradient this step
        self.arrived = [False] * len(params)   # its accumulation node ran (possibly with an undefined gradient)
        self.launched = False


class GradReducer:
    """Bucketed, backward-overlapped all-reduce (mean) of the trainable gradients.

        reducer = GradReducer(model.parameters(), bucket_mb=16)
        reducer.prepare(); loss.backward(); reducer.finish()   # .grad now holds the cross-rank mean
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 16.0,
                 process_group: Optional[dist.ProcessGroup] = None, force: bool = False,
                 buffers: Optional[Iterable[torch.Tensor]] = None, broadcast: bool = True, static_graph: bool = True):
        """force=True runs the pack / all-reduce / unpack path even on a single rank (hardware test of the RCCL path).
        broadcast: every parameter handed over (frozen ones included) and every tensor of ``buffers`` is overwritten with
        rank 0's copy, so replicas start identical whatever each rank's seed or checkpoint did (DDP's constructor
        semantics); static_graph: the set of parameters that receive a gradient is the same every step (true for this
        model: it is structural), so ranks agree on it once, at the first step, instead of every step."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.static_graph = static_graph
        self._global_seen: Optional[List[List[bool]]] = None
        all_params = list(params)
        if broadcast and self.active:
            # p.detach(), not p.data: the copy must bump the Parameter's version counter -- a layer that has already run a forward
            # holds compute-dtype copies of its frozen weights keyed by that version (lora.MTLoRALinear._weights) and would keep
            # multiplying with the pre-broadcast values (found by the two-rank test of round 6, where rank 1 is seeded differently)
            self._broadcast_from_rank0([p.detach() for p in all_params] + ([] if buffers is None else list(buffers)))
        plist = [p for p in all_params if p.requires_grad]
        plist.reverse()  # approximate backward completion order
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in plist:
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(_Bucket(cur))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._armed = False
        # streams other than the ambient one on which gradients may be produced (a model that runs independent branches
        # on side streams): a bucket is packed from an autograd hook whose current stream only orders ITS OWN gradient,
        # so the pack first waits for these
        self.extra_streams: List["torch.cuda.Stream"] = []

    def _broadcast_from_rank0(self, tensors: List[torch.Tensor], chunk_bytes: int = 64 << 20) -> None:
        """rank 0's values into every rank's tensors: coalesced by dtype into flat chunks (a few large broadcasts, not one
        per tensor -- the C2 model has 431 state tensors)."""
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        by_dtype = {}
        for t in tensors:
            if t.numel():
                by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for (_, _), ts in by_dtype.items():
            cur, cur_b = [], 0
            groups = []
            for t in ts:
                nb = t.numel() * t.element_size()
                if cur and cur_b + nb > chunk_bytes:
                    groups.append(cur)
                    cur, cur_b = [], 0
                cur.append(t)
                cur_b += nb
            if cur:
                groups.append(cur)
            for g in groups:
                flat = torch.cat([t.reshape(-1) for t in g])
                dist.broadcast(flat, src=src, group=self.group)
                torch._foreach_copy_(g, [v.view(t.shape) for v, t in zip(flat.split([t.numel() for t in g]), g)])

    def _agree_on_used(self) -> None:
        """which parameters got a gradient on ANY rank this step (one small MAX all-reduce + one host read).  With
        ``static_graph`` this runs once; a parameter used elsewhere but not here then receives the mean (its own
        contribution being zero) instead of silently keeping ``grad None`` while other replicas step it."""
        flags = torch.tensor([1.0 if s else 0.0 for b in self.buckets for s in b.seen], dtype=torch.float32,
                             device=self.buckets[0].flat.device)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        f = flags.tolist()
        out, i = [], 0
        for b in self.buckets:
            out.append([v > 0.0 for v in f[i:i + len(b.params)]])
            i += len(b.params)
        self._global_seen = out

    @property
    def nbytes(self) -> int:
        return sum(b.flat.numel() * 4 for b in self.buckets)

    def prepare(self, defer: bool = False) -> None:
        """call before backward.  defer=True (graphed step): buckets are only PACKED as their gradients arrive (those
        copies are captured with the backward graph); the collectives are issued afterwards by ``all_reduce_packed``
        outside any graph, and ``unpack`` (capturable) writes the means back."""
        self._defer = defer
        self._ambient = (torch.cuda.current_stream(self.buckets[0].flat.device)
                         if self.buckets and self.buckets[0].flat.is_cuda else None)
        for b in self.buckets:
            b.pending = len(b.params)
            b.handle = None
            b.seen = [False] * len(b.params)
            b.arrived = [False] * len(b.params)
            b.launched = False
        self._next = 0  # buckets are launched strictly in index order (see _launch_ready)
        self._armed = True

    def _pack(self, b: _Bucket) -> None:
        """gather the bucket's gradients into its flat buffer: ONE multi-tensor copy (not one kernel per parameter)."""
        if self.extra_streams and b.flat.is_cuda:
            cur = torch.cuda.current_stream(b.flat.device)
            for st in list(self.extra_streams) + ([self._ambient] if getattr(self, "_ambient", None) is not None else []):
                if st != cur:
                    cur.wait_stream(st)
        dst, src = [], []
        for pi, p in enumerate(b.params):
            view = b.flat[b.offsets[pi]:b.offsets[pi] + p.numel()]
            if b.seen[pi]:
                dst.append(view)
                src.append(p.grad.reshape(-1))
            else:  # no gradient on this rank this step: counts as zero
                view.zero_()
        if dst:
            torch._foreach_copy_(dst, src)

    def _launch(self, b: _Bucket) -> None:
        self._pack(b)
        b.launched = True
        if self.active and not getattr(self, "_defer", False):
            b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch_ready(self, flush: bool = False) -> None:
        """issue the collectives in BUCKET ORDER, whatever order the buckets complete in: RCCL / NCCL match collectives by issue
        order, and a rank on which one parameter of bucket i gets no gradient this step would otherwise issue bucket i from
        ``finish`` -- after buckets the other ranks issued later -- and hang or sum mismatched buckets.  A complete bucket
        behind an incomplete one waits for it (``flush``: backward is over, incomplete buckets count their missing gradients
        as zero).  Buckets follow the reverse parameter order, i.e. roughly the order backward completes them in."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.pending > 0:
                if not flush:
                    return
                b.pending = 0
            if not b.launched:
                self._launch(b)
            self._next += 1

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._armed:
            return
        bi, pi = self._where[p]
        b = self.buckets[bi]
        if b.arrived[pi]:
            return  # gradient accumulation touching the same parameter twice in one backward
        b.arrived[pi] = True
        # the hook also fires when the incoming gradient is undefined (an unused output of a custom Function, e.g. the
        # final stage's shared fc2 factors): such a parameter has arrived but contributes zeros and keeps grad None
        b.seen[pi] = p.grad is not None
        b.pending -= 1
        if b.pending == 0:
            self._launch_ready()

    def finish(self) -> None:
        """call after backward: flush incomplete buckets, wait, write the mean back into ``.grad``."""
        self._armed = False
        self._launch_ready(flush=True)  # parameters that produced no gradient on this rank this step count as zero
        if self.active and self.buckets and (self._global_seen is None or not self.static_graph):
            self._agree_on_used()
        inv = 1.0 / self.world
        for bi, b in enumerate(self.buckets):
            if b.handle is not None:
                b.handle.wait()
            if self.active:
                b.flat.mul_(inv)
            dst, src = [], []
            for pi, p in enumerate(b.params):
                used = b.seen[pi] or (self.active and self._global_seen is not None and self._global_seen[bi][pi])
                if not used:  # no rank produced a gradient: grad stays None everywhere (the optimizer skips it everywhere)
                    continue
                if p.grad is None:  # used on another rank only: this rank's share is zero, the mean is still its gradient
                    p.grad = torch.empty_like(p)
                dst.append(p.grad)
                src.append(b.flat[b.offsets[pi]:b.offsets[pi] + p.numel()].view_as(p.grad))
            if dst and self.active:
                torch._foreach_copy_(dst, src)

    # ---- graphed step: backward graph (packs) -> all_reduce_packed (eager RCCL) -> optimizer graph (unpack first)
    def flush_packs(self) -> None:
        """after backward, inside the backward capture: pack the buckets whose last gradient never arrived."""
        self._armed = False
        self._launch_ready(flush=True)  # (deferred mode: "launch" = pack; also the complete buckets held back behind an incomplete one)

    def all_reduce_packed(self) -> None:
        if self.active:
            for b in self.buckets:
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)

    def unpack(self) -> None:
        """capturable: mean = sum / world, written back into the gradients that exist."""
        if not self.active:
            return
        inv = 1.0 / self.world
        for b in self.buckets:
            b.flat.mul_(inv)
            dst, src = [], []
            for pi, p in enumerate(b.params):
                if b.seen[pi] and p.grad is not None:
                    dst.append(p.grad)
                    src.append(b.flat[b.offsets[pi]:b.offsets[pi] + p.numel()].view_as(p.grad))
            if dst:
                torch._foreach_copy_(dst, src)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
