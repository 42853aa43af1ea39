"""torch.autograd.Function wrappers over the C ABI (include/mtlora_hip.h).

Host-side plumbing only: allocate outputs / ctx / scratch with torch, pass raw pointers and the
current stream to the library.  Every op raises if the library or a GPU tensor is missing.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

# ----------------------------------------------------------------------------------------------
# dropout seeds: one fresh 64-bit seed per MTLoRALinear call in training mode, derived from
# torch.initial_seed() and a call counter (deterministic under torch.manual_seed + fixed call order)
# ----------------------------------------------------------------------------------------------
_seed_counter = 0


# optional device-resident offset added to every dropout seed when the kernels RUN (mtlora_linear_desc.seed_offset):
# a captured HIP graph bakes the host-side seeds in, so a graphed train step bumps this one int64 between replays.
_seed_offset: Optional[torch.Tensor] = None


def set_seed_offset(t: Optional[torch.Tensor]) -> None:
    """install (or clear) the device int64[1] tensor whose value is added to the dropout seeds at kernel time."""
    global _seed_offset
    if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
        raise RuntimeError("mtlora_amd: the seed offset must be a CUDA int64 tensor with one element")
    _seed_offset = t


# optional second stream for the factor gradients (dA_*, dB_*: the split-M k_tn reductions) of MTLoRALinear.backward:
# nothing in the backward chain depends on them (only the optimizer / the gradient reducer does), and they are a fifth of
# the linear path's kernel time, mostly as latency-bound launches -- on their own stream they fill the gaps of the main
# chain.  The caller that installs a stream must JOIN it before the gradients are read (train_step does; a GradReducer gets
# it through ``extra_streams``).  Off (None) by default: a plain ``loss.backward()`` stays single-stream.
_factor_stream: Optional["torch.cuda.Stream"] = None
# layers with fewer rows than this keep their factor gradients on the main stream: the fork / join costs ~30 us of host time per
# layer and only pays while the kernels are long enough for the GPU to be the bottleneck (C1 at batch 2, M <= 6272 everywhere,
# ran 35 % SLOWER with the side stream: 18.7 vs 13.8 ms / step)
_FACTOR_MIN_M = int(os.environ.get("MTLORA_FACTOR_MIN_M", "8192"))


# factor parameters whose gradient was written on the side stream by an earlier backward call -> (that stream, the autograd graph
# task it happened in).  A later use of the same layer makes its own stream WAIT for the side stream before anything touches the
# gradient; whether the side stream may be used again depends on WHEN the entry was written:
#   * in the CURRENT backward pass (the layer is applied twice in one graph): the autograd engine sums the two gradients of a factor
#     on the stream of the forward call, so the second gradient stays on that stream;
#   * in an EARLIER pass that nobody joined: once waited for, the side stream is free again (unless the factor's .grad exists, in
#     which case autograd accumulates into it on this stream -- ``use_side`` checks that separately).
# Keys are compared by IDENTITY (a WeakKeyDictionary compares its referents with ``==``, which is element-wise for
# tensors -- looking up a present key raised "Boolean value of Tensor ... is ambiguous"); weak, so a freed Parameter leaves no
# entry behind (raw ids can be reused by other objects).  A caller that joins the side stream AFTER the backward pass (train_step,
# GraphedTrainStep) says so with ``factor_stream_joined()``; the GradReducer's mid-pass waits do not (the same-pass rule above
# must keep holding for the rest of the pass).
from torch.utils.weak import WeakIdKeyDictionary  # noqa: E402

_side_pending: "WeakIdKeyDictionary" = WeakIdKeyDictionary()


def factor_stream_joined() -> None:
    """the caller made its stream wait for the side stream: nothing written there is pending any more"""
    _side_pending.clear()


def set_factor_stream(stream: Optional["torch.cuda.Stream"]) -> None:
    """install (or, with None, remove) the side stream for the factor gradients (see above).  Entries of ``_side_pending`` stay:
    whoever installs / removes the stream has not necessarily joined the previous one (``factor_stream_joined`` says that)."""
    global _factor_stream
    _factor_stream = stream


def _graph_task_id() -> int:
    return torch._C._current_graph_task_id()


def _side_join_pending(params, dev) -> bool:
    """make the current stream wait for every side stream that still holds a gradient of ``params`` (the layer's factor
    Parameters) from an earlier backward call.  Returns True when one of those entries was written in the CURRENT backward pass
    (same layer twice in one graph): this call's factor gradients must then stay on the current stream."""
    if not _side_pending:
        return False
    same_pass, streams, task = False, {}, _graph_task_id()
    for p in params:
        if p is None:
            continue
        ent = _side_pending.pop(p, None)
        if ent is not None:
            streams[id(ent[0])] = ent[0]
            same_pass = same_pass or ent[1] == task
    cur = torch.cuda.current_stream(dev) if streams else None
    for st in streams.values():
        cur.wait_stream(st)
    if same_pass:
        # a THIRD (fourth ...) use of the layer in this pass must see the same-pass rule too -- the entries popped above are
        # put back under the current graph task, so every later use keeps its factor gradients on the stream autograd sums them on
        side = next(iter(streams.values()))
        for p in params:
            if p is not None:
                _side_pending[p] = (side, task)
    return same_pass


def _side_safe_params(params) -> bool:
    """may the gradients of ``params`` be WRITTEN on the side stream?  Only when autograd will take the gradient tensor as it is:
    no existing .grad to accumulate into (an add on the backward's stream), no tensor hook on the Parameter (a hook receives --
    and may replace, i.e. force a copy of -- the gradient on the backward's stream while the side stream is still writing it) and
    no NON-LEAF factor (lora.py's rank-aware Delta W: its producer's backward reads the gradient on the backward's stream)."""
    return all(p is None or (p.grad is None and not p._backward_hooks and (p.is_leaf or not p.requires_grad)) for p in params)


def _side_mark_pending(params, side) -> None:
    task = _graph_task_id()
    for p in params:
        if p is not None:
            _side_pending[p] = (side, task)


def factor_stream() -> Optional["torch.cuda.Stream"]:
    return _factor_stream


# ----------------------------------------------------------------------------------------------
# DropPath keep / scale vectors, drawn in bulk.  Every residual of the backbone needs n x B per-sample factors
# bernoulli(keep) / keep (timm DropPath, swin_transformer_mtlora.py:388-426): 22 draws per Swin-T step = 44 tiny launches
# (bernoulli_, div_) strung along the forward's critical path.  A train step can announce itself (``droppath_begin_step`` /
# ``droppath_end_step``, mtl_harness.train_step): the sequence of requests (n, B, keep) seen in one step is replayed as the PLAN of
# the next one, whose factors then come from ONE rand + compare + scale over all of them; a request that deviates from the plan
# (other model, other batch, eval) simply draws on its own and the plan is re-recorded.  Same distribution, same global CUDA
# generator; the individual values differ from per-call draws.
# ----------------------------------------------------------------------------------------------
class _DropPathPool:
    def __init__(self):
        self.plan, self.rec, self.rows, self.at, self.active, self.in_step = [], [], None, 0, False, False
        self.cache_key, self.keep_col = None, None

    def begin(self, device):
        self.rec, self.at, self.in_step = [], 0, True
        capturing = torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()
        self.active = bool(self.plan) and not capturing
        if not self.active:
            return
        key = (tuple(self.plan), str(device))
        if key != self.cache_key:  # per-row keep probabilities of the plan (built once)
            B = self.plan[0][1]
            if any(b != B for _, b, _ in self.plan):
                self.active = False
                return
            keeps = [k for n, _, k in self.plan for _ in range(n)]
            self.keep_col = torch.tensor(keeps, dtype=torch.float32, device=device).unsqueeze(1)
            self.cache_key = key
        B = self.plan[0][1]
        u = torch.rand(self.keep_col.shape[0], B, dtype=torch.float32, device=device)
        self.rows = (u < self.keep_col) / self.keep_col  # bernoulli(keep) / keep, all requests of the step at once
        self.off = 0

    def end(self):
        if self.in_step:
            self.plan, self.in_step, self.active, self.rows = self.rec, False, False, None

    def scale(self, n: int, B: int, keep: float, device) -> torch.Tensor:
        if self.in_step:
            self.rec.append((n, B, keep))
            if self.active and self.at < len(self.plan) and self.plan[self.at] == (n, B, keep) and self.rows.device == device:
                out = self.rows[self.off:self.off + n]
                self.off += n
                self.at += 1
                return out
            self.active = False
        return torch.empty(n, B, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)


_droppath_pool = _DropPathPool()


def droppath_begin_step(device) -> None:
    _droppath_pool.begin(device)


def droppath_end_step() -> None:
    _droppath_pool.end()


def droppath_reset() -> None:
    """forget the recorded plan: the next announced step draws per request again (two runs that must see the same random
    stream have to start from the same pool state)."""
    global _droppath_pool
    _droppath_pool = _DropPathPool()


def droppath_scale(n: int, B: int, keep: float, device) -> torch.Tensor:
    """(n, B) fp32 factors bernoulli(keep) / keep -- from the step's bulk draw when one is active (see _DropPathPool)."""
    return _droppath_pool.scale(n, B, keep, device)


# ----------------------------------------------------------------------------------------------
# kernel selection (mtlora_linear_desc.sel_* / max_cu, ABI v6).  The LIBRARY reads no environment variables; this module reads
# the developer switches once at import and hands them to every call through the descriptor.  ``set_tuning`` overrides them in
# process (tests: the "[tiled]" / "[dense]" / "[persist]" families of tests/conftest.py; tools/bench_linear.py A/B runs).
#   MTLORA_SP=0        -> stream=1  (tiled kernels only)         MTLORA_NTD=0 / 2      -> dense=1 / 2 (k_ntd never / whenever)
#   MTLORA_SP_TN=0 / 2 -> tn=1 / 2  (k_sp_tn never / whenever)   MTLORA_SP_PROJK=0 / 2 -> projk=1 / 2
#   MTLORA_MAX_CU=n    -> max_cu=n  (persistent grids sized as if the device had n CUs)
# ----------------------------------------------------------------------------------------------
def _env_tri(name: str) -> int:
    v = os.environ.get(name)
    return 0 if v is None or v == "1" else (1 if v == "0" else int(v))


_TUNING_DEFAULT = {"stream": 1 if os.environ.get("MTLORA_SP") == "0" else 0, "dense": _env_tri("MTLORA_NTD"),
                   "tn": _env_tri("MTLORA_SP_TN"), "projk": _env_tri("MTLORA_SP_PROJK"),
                   "max_cu": int(os.environ.get("MTLORA_MAX_CU", "0"))}
_tuning = dict(_TUNING_DEFAULT)


def set_tuning(**kw) -> dict:
    """override kernel-selection switches for the calls that follow (keys: stream, dense, tn, projk, max_cu; values as in
    include/mtlora_hip.h: 0 = library default); ``set_tuning()`` with no arguments restores the import-time values.
    Returns the previous settings."""
    prev = dict(_tuning)
    if not kw:
        _tuning.update(_TUNING_DEFAULT)
    for k, v in kw.items():
        if k not in _tuning:
            raise KeyError(f"mtlora_amd: unknown tuning key {k!r} (have {sorted(_tuning)})")
        _tuning[k] = int(v)
    return prev


def next_seed() -> int:
    global _seed_counter
    _seed_counter += 1
    x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    return x & 0xFFFFFFFFFFFFFFFF


_HOT_DTYPES = (torch.float32, torch.bfloat16, torch.float16)
_GLUE_DTYPES = (torch.float32, torch.bfloat16, torch.float16)  # LayerNorm family, residual + DropPath, BatchNorm + ReLU kernels
_DT_CODE = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}


def compute_dtype(x: torch.Tensor) -> torch.dtype:
    """dtype the hot path (MTLoRALinear, window attention) runs in: the autocast dtype when autocast is on (like F.linear),
    else x's.  bf16 is the dtype to use on MI355X (same MFMA rate as fp16, no loss scaling); fp16 -- the reference's
    ``torch.cuda.amp.autocast()`` default, main.py:341 -- is supported by the linear / attention kernels, with the block glue
    (LayerNorm, residuals, heads) falling back to fp32 outputs / ATen for it."""
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        if dt not in _HOT_DTYPES:
            raise RuntimeError(f"mtlora_amd: unsupported autocast dtype {dt}")
        return dt
    if x.dtype not in _HOT_DTYPES:
        raise RuntimeError(f"mtlora_amd: unsupported activation dtype {x.dtype}")
    return x.dtype


def glue_dtype(x: torch.Tensor) -> torch.dtype:
    """output dtype of the fused glue kernels in front of an MTLoRALinear: the compute dtype (fp32, or bf16 / fp16 under
    autocast -- the LayerNorm-family, residual and BatchNorm kernels are instantiated for all three)."""
    return compute_dtype(x)


# ----------------------------------------------------------------------------------------------
# MTLoRALinear
# ----------------------------------------------------------------------------------------------
@dataclass
class LinearMeta:
    """static description of one MTLoRALinear call (lora.py:253-284)."""
    K: int
    N: int
    r_s: int
    r_t: Tuple[int, ...]
    scale_s: float
    scale_t: Tuple[float, ...]
    mode: int                 # 0 matrix, 1 matrixv2
    has_x_tasks: bool
    dropout_p: float          # 0 in eval
    seed: int
    dtype: torch.dtype
    weight_requires_grad: bool = False
    n_scale_t: int = 0        # >0: per-task scales are trainable Parameters passed after B_t
    n_gate: int = 0           # >0: x (and x_t) = gelu(gate): the LAST n_gate args are the pre-activations; dx *= gelu'(gate)
    gelu_out: bool = False    # also return gelu(y) for every output (fc1 of the Mlp): outputs = (y_s, *y_t, a_s, *a_t)
    packed: Optional[torch.Tensor] = None  # the layer's factors packed ahead of time (PackTable / mtlora_linear_pack): fwd skips k_pack

    @property
    def T(self) -> int:
        return len(self.r_t)

    def desc(self, M: int) -> L.LinearDesc:
        d = L.LinearDesc()
        d.M, d.K, d.N = M, self.K, self.N
        d.dtype = _DT_CODE[self.dtype]
        d.mode, d.T, d.r_s = self.mode, self.T, self.r_s
        for i, r in enumerate(self.r_t):
            d.r_t[i] = r
            d.scale_t[i] = self.scale_t[i]
        d.scale_s = self.scale_s
        d.has_x_tasks = 1 if self.has_x_tasks else 0
        d.dropout_p = self.dropout_p
        d.seed = self.seed
        d.seed_offset = 0 if _seed_offset is None else _seed_offset.data_ptr()
        tu = _tuning
        d.sel_stream, d.sel_dense, d.sel_tn, d.sel_projk, d.max_cu = tu["stream"], tu["dense"], tu["tn"], tu["projk"], tu["max_cu"]
        d.packed = 0 if self.packed is None else self.packed.data_ptr()
        return d


def _as2d(t: torch.Tensor, cols: int, dtype: torch.dtype) -> torch.Tensor:
    """t viewed as a contiguous (-1, cols) matrix of ``dtype`` (no op at all in the common case)."""
    if t.dtype == dtype and t.is_contiguous():
        return t.view(-1, cols)
    return t.reshape(-1, cols).to(dtype).contiguous()


def _flat(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """t itself when it is already contiguous in ``dtype`` (the kernels take a pointer and a row count: no 2-D view object is needed --
    each ``.view`` costs ~1.5 us of dispatch, 10 of them per MTLoRALinear call), else a contiguous copy in ``dtype``"""
    if t.dtype == dtype and t.is_contiguous():
        return t
    return t.to(dtype).contiguous()


def _f32c(p):
    """fp32 contiguous view of a factor (Parameters already are: used as they come, autograd is off inside Function.forward)."""
    if p is None or (p.dtype == torch.float32 and p.is_contiguous()):
        return p
    return p.detach().float().contiguous()


_bytes_cache: dict = {}  # (kind, M, K, N, r_s, r_t, T-independent fields ...) -> ctx / scratch bytes (one ctypes call per new shape)


def _desc_bytes(kind: str, fn, meta: "LinearMeta", M: int, d) -> int:
    key = (kind, M, meta.K, meta.N, meta.r_s, meta.r_t, meta.has_x_tasks, meta.dtype, meta.mode, meta.packed is not None)
    v = _bytes_cache.get(key)
    if v is None:
        v = fn(ctypes.byref(d))
        _bytes_cache[key] = v
    return v


class PackTable:
    """ONE launch per optimizer step that packs the low-rank factors of many MTLoRALinear layers (``mtlora_linear_pack_table``).

    ``entries``: list of (meta, A_s, B_s, A_t list, B_t list, packed uint8 tensor) -- ``meta`` a LinearMeta carrying the scales,
    dropout_p and has_x_tasks the layer is CALLED with (they enter the alpha-scaled copies).  The table holds raw device pointers of
    the masters and the destinations: valid while those tensors live and are only updated in place (``nn.Parameter``s under an
    optimizer; the packed buffers are owned by the layers)."""

    def __init__(self, entries, device, dtype: torch.dtype):
        lib = L.lib()
        eb = lib.mtlora_linear_pack_entry_bytes()
        host = (ctypes.c_ubyte * (eb * max(len(entries), 1)))()
        self.keep = entries  # (keeps every tensor whose address is in the table alive)
        for i, (meta, A_s, B_s, A_t, B_t, packed, *_owner) in enumerate(entries):
            d = meta.desc(1)
            d.packed = 0
            st = lib.mtlora_linear_pack_entry(ctypes.byref(d), L.ptr(A_s), L.ptr(B_s), L.ptr_array(A_t), L.ptr_array(B_t), L.ptr(packed),
                                              packed.numel(), ctypes.byref(host, i * eb))
            L.check(st, "mtlora_linear_pack_entry")
        self.n = len(entries)
        self.dtype_code = _DT_CODE[dtype]
        self.table = torch.frombuffer(host, dtype=torch.uint8).clone().to(device)

    def pack(self) -> None:
        if self.n:
            L.check(L.lib().mtlora_linear_pack_table(L.ptr(self.table), self.n, self.dtype_code, L.stream_ptr()), "mtlora_linear_pack_table")


def packed_bytes(meta: "LinearMeta") -> int:
    d = meta.desc(1)
    d.packed = 0
    n = L.lib().mtlora_linear_packed_bytes(ctypes.byref(d))
    if n < 0:
        raise RuntimeError("mtlora_amd: invalid MTLoRALinear shape for mtlora_linear_packed_bytes")
    return n


class MTLoRALinearFn(torch.autograd.Function):
    """(y_s, y_t[0..T-1]) = f(x, x_t[0..T-1], A_s, B_s, A_t[..], B_t[..]); W (and bias) frozen by default.

    args: meta, x, W_c, Wt_c, bias_f32, W_master, bias_master, A_s, B_s, scale_s_param,
          *x_t(T or 0), *A_t(T), *B_t(T), *scale_t_params(T or 0), *gelu_gates(0 or 1 + len(x_t))
    W_c / Wt_c are the compute-dtype copies the module caches; W_master / bias_master are only used to
    route gradients when the pretrained weight is left trainable (MTLORA.FREEZE_PRETRAINED False);
    scale_*_param are the 1-element Parameters of TRAINABLE_SCALE_* (None otherwise): the kernels take the
    scales as scalars and d(loss)/d(scale) = <dB, B> / scale is formed from the factor gradient."""

    @staticmethod
    def forward(ctx, meta: LinearMeta, x, W_c, Wt_c, bias_f32, W_master, bias_master, A_s, B_s, scale_s_param, *rest):
        ctx.set_materialize_grads(False)  # an unused output must reach backward as None, not as zeros
        T = meta.T
        nx = T if meta.has_x_tasks else 0
        x_t, A_t, B_t = list(rest[:nx]), list(rest[nx:nx + T]), list(rest[nx + T:nx + 2 * T])
        L.require_gpu(x, W_c, *x_t)
        lead = x.shape[:-1]
        x2 = _flat(x, meta.dtype)              # (M x K) row-major whatever the leading shape
        xt2 = [_flat(t, meta.dtype) for t in x_t]
        M = x2.numel() // meta.K
        if x2.shape[-1] != meta.K or any(t.shape != x2.shape for t in xt2):
            raise RuntimeError(f"mtlora_amd: MTLoRALinear input of shape {tuple(x.shape)} for in_features = {meta.K}")
        d = meta.desc(M)
        lib = L.lib()
        ctx_bytes = _desc_bytes("ctx", lib.mtlora_linear_ctx_bytes, meta, M, d)
        if ctx_bytes < 0:
            raise RuntimeError(f"mtlora_amd: invalid MTLoRALinear shape M={M} K={meta.K} N={meta.N} (K, N must be "
                               "multiples of 8)")
        ctxbuf = torch.empty(ctx_bytes, dtype=torch.uint8, device=x.device)
        oshape = (*lead, meta.N)
        ys = torch.empty(oshape, dtype=meta.dtype, device=x.device)
        yt = [torch.empty(oshape, dtype=meta.dtype, device=x.device) for _ in range(T)]
        A_s_c, B_s_c = _f32c(A_s), _f32c(B_s)
        A_t_c, B_t_c = [_f32c(a) for a in A_t], [_f32c(b) for b in B_t]
        acts = []
        if meta.gelu_out:  # second outputs gelu(y) written by the same epilogue (mtlora_linear_fwd_gelu)
            acts = [torch.empty(oshape, dtype=meta.dtype, device=x.device) for _ in range(1 + T)]
            st = lib.mtlora_linear_fwd_gelu(ctypes.byref(d), L.ptr(x2), L.ptr_array(xt2), L.ptr(W_c), L.ptr(bias_f32),
                                            L.ptr(A_s_c), L.ptr(B_s_c), L.ptr_array(A_t_c), L.ptr_array(B_t_c), L.ptr(ys),
                                            L.ptr_array(yt), L.ptr(acts[0]), L.ptr_array(acts[1:]), L.ptr(ctxbuf), ctx_bytes,
                                            L.stream_ptr())
            L.check(st, "mtlora_linear_fwd_gelu")
        else:
            st = lib.mtlora_linear_fwd(ctypes.byref(d), L.ptr(x2), L.ptr_array(xt2), L.ptr(W_c), L.ptr(bias_f32),
                                       L.ptr(A_s_c), L.ptr(B_s_c), L.ptr_array(A_t_c), L.ptr_array(B_t_c), L.ptr(ys),
                                       L.ptr_array(yt), L.ptr(ctxbuf), ctx_bytes, L.stream_ptr())
            L.check(st, "mtlora_linear_fwd")
        ctx.meta, ctx.lead, ctx.nx = meta, lead, nx
        ctx.in_dtypes = [x.dtype] + [t.dtype for t in x_t]
        gates = []
        if meta.n_gate:
            gates = list(rest[len(rest) - meta.n_gate:])
            if len(gates) != 1 + nx or any(g.dtype != meta.dtype or not g.is_contiguous() or g.numel() != x2.numel() or g.shape[-1] != meta.K
                                           for g in gates):
                raise RuntimeError("mtlora_amd: gelu gates must be contiguous pre-activations of x / x_t in the compute dtype")
        ctx.save_for_backward(x2, Wt_c, ctxbuf, *xt2, *gates)
        ctx.keep = (A_s_c, B_s_c, A_t_c, B_t_c)  # fp32 factor views (also used for the trainable-scale gradients)
        # (meta.packed keeps the packed-factor buffer alive until backward; a trainer overwrites it only after the step's backward)
        ctx.factor_params = (A_s, B_s, *A_t, *B_t)  # the Parameters themselves: backward looks at their .grad (side stream)
        ctx.has_scale_s = scale_s_param is not None
        ctx.M = M
        return (ys, *yt, *acts)

    @staticmethod
    def backward(ctx, *grads):
        meta: LinearMeta = ctx.meta
        T, nx = meta.T, ctx.nx
        if meta.gelu_out:
            # a gradient arriving on a_k = gelu(y_k) comes from a consumer that already applied gelu'(y_k) (an MTLoRALinear
            # called with gelu_gate = y_k): it IS a gradient w.r.t. y_k
            gy, ga = grads[:1 + T], grads[1 + T:]
            grads = tuple(a if y is None else (y if a is None else y + a) for y, a in zip(gy, ga))
        if all(g is None for g in grads):
            return (None,) * (10 + nx + 2 * T + meta.n_scale_t + meta.n_gate)
        x2, Wt_c, ctxbuf, *xt2 = ctx.saved_tensors
        gates = []
        if meta.n_gate:
            xt2, gates = xt2[:nx], xt2[nx:]
        M = ctx.M
        dev = x2.device
        g2 = [None if g is None else _flat(g, meta.dtype) for g in grads]
        dy_s, dy_t = g2[0], g2[1:1 + T]
        d = meta.desc(M)
        lib = L.lib()
        scratch_bytes = _desc_bytes("bwd", lib.mtlora_linear_bwd_scratch_bytes, meta, M, d)
        scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
        need = ctx.needs_input_grad  # (meta, x, W_c, Wt_c, bias_f32, W_master, bias_master, A_s, B_s, scale_s, *rest)
        ishape = (*ctx.lead, meta.K)
        dx = torch.empty(ishape, dtype=meta.dtype, device=dev)
        dxt = [torch.empty(ishape, dtype=meta.dtype, device=dev) for _ in range(nx)]
        has_s = meta.r_s > 0 and (dy_s is not None or (meta.mode == 1 and any(g is not None for g in dy_t)))
        # (a factor that takes no gradient -- the identity side of a rank-aware update, a frozen factor -- is not reduced at all: the
        # library skips the problems whose output pointer is null)
        nA, nB = 10 + nx, 10 + nx + T
        dA_s = torch.empty((meta.r_s, meta.K), dtype=torch.float32, device=dev) if (has_s and need[7]) else None
        dB_s = torch.empty((meta.N, meta.r_s), dtype=torch.float32, device=dev) if (has_s and need[8]) else None
        dA_t = [torch.empty((meta.r_t[t], meta.K), dtype=torch.float32, device=dev) if (dy_t[t] is not None and need[nA + t]) else None
                for t in range(T)]
        dB_t = [torch.empty((meta.N, meta.r_t[t]), dtype=torch.float32, device=dev) if (dy_t[t] is not None and need[nB + t]) else None
                for t in range(T)]
        if nx:  # a task input whose output got no gradient still needs a defined (zero) gradient
            for t in range(T):
                if dy_t[t] is None:
                    dxt[t].zero_()
        def launch(stream_ptr):
            if gates:  # the inputs were gelu(gate): the dX epilogue also applies gelu'(gate) (GELU backward fused)
                st = lib.mtlora_linear_bwd_gelu(ctypes.byref(d), L.ptr(x2), L.ptr_array(xt2), L.ptr(Wt_c), L.ptr(dy_s),
                                                L.ptr_array(dy_t), L.ptr(ctxbuf), ctxbuf.numel(), L.ptr(dx), L.ptr_array(dxt),
                                                L.ptr(dA_s), L.ptr(dB_s), L.ptr_array(dA_t), L.ptr_array(dB_t), L.ptr(scratch),
                                                scratch_bytes, L.ptr(gates[0]), L.ptr_array(gates[1:]), stream_ptr)
                L.check(st, "mtlora_linear_bwd_gelu")
            else:
                st = lib.mtlora_linear_bwd(ctypes.byref(d), L.ptr(x2), L.ptr_array(xt2), L.ptr(Wt_c), L.ptr(dy_s),
                                           L.ptr_array(dy_t), L.ptr(ctxbuf), ctxbuf.numel(), L.ptr(dx), L.ptr_array(dxt),
                                           L.ptr(dA_s), L.ptr(dB_s), L.ptr_array(dA_t), L.ptr_array(dB_t), L.ptr(scratch),
                                           scratch_bytes, stream_ptr)
                L.check(st, "mtlora_linear_bwd")

        side = _factor_stream
        fgrads = [t for t in [dA_s, dB_s, *dA_t, *dB_t] if t is not None]
        # the factor gradients go to the side stream only when nothing on this stream reads them inside backward: no
        # trainable-scale gradient (formed from dB below) and no gradient accumulation into an existing .grad
        use_side = (side is not None and fgrads and not ctx.has_scale_s and meta.n_scale_t == 0
                    and side.device == dev and M >= _FACTOR_MIN_M and _side_safe_params(ctx.factor_params))
        if _side_join_pending(ctx.factor_params, dev):
            use_side = False  # same layer earlier in THIS backward pass: autograd sums the two gradients on this stream
        if use_side:
            d.bwd_phase = 1
            launch(L.stream_ptr())
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            d.bwd_phase = 2
            launch(ctypes.c_void_p(side.cuda_stream))
            _side_mark_pending(ctx.factor_params, side)
            for t in [x2, ctxbuf, scratch, *xt2, *fgrads] + [g for g in g2 if g is not None]:
                t.record_stream(side)  # allocated on this stream, still in use on the side stream when freed here
        else:
            launch(L.stream_ptr())
        dW = dbias = None
        if need[5] or need[6]:
            # pretrained weight and / or bias left trainable (MTLORA.FREEZE_PRETRAINED False; mark_only_lora_as_trainable
            # bias='all' trains `linear.bias` with W frozen, reference lora.py:606-617): dense gradients outside the frozen-W
            # hot path.  Every output adds the same base  x W^T + b, so both see G = sum of the output gradients.
            G = None
            for g in g2:
                if g is not None:
                    G = g.reshape(M, meta.N).float() if G is None else G + g.reshape(M, meta.N).float()
            if G is not None:
                if need[5]:
                    dW = G.t() @ x2.reshape(M, meta.K).float()
                if need[6]:
                    dbias = G.sum(0)
        dxo = dx if ctx.in_dtypes[0] == meta.dtype else dx.to(ctx.in_dtypes[0])
        dxto = [dxt[t] if ctx.in_dtypes[1 + t] == meta.dtype else dxt[t].to(ctx.in_dtypes[1 + t]) for t in range(nx)]
        _, B_s_c, _, B_t_c = ctx.keep
        d_ss = None
        if ctx.has_scale_s and dB_s is not None and meta.scale_s != 0.0:
            d_ss = ((dB_s * B_s_c).sum() / meta.scale_s).reshape(1)
        d_st = []
        for t in range(meta.n_scale_t):
            ok = dB_t[t] is not None and meta.scale_t[t] != 0.0
            d_st.append(((dB_t[t] * B_t_c[t]).sum() / meta.scale_t[t]).reshape(1) if ok else None)
        return (None, dxo, None, None, None, dW, dbias, dA_s, dB_s, d_ss, *dxto, *dA_t, *dB_t, *d_st, *([None] * meta.n_gate))


# ----------------------------------------------------------------------------------------------
# Task-enabled Mlp with IMPLICIT task hidden tensors (csrc/hid.hip, ABI v8)
# ----------------------------------------------------------------------------------------------
_MLP_HID = os.environ.get("MTLORA_MLP_HID", "1") != "0"
_hid_ok_cache: dict = {}


def mlp_hid_enabled() -> bool:
    return _MLP_HID


def set_mlp_hid(on: bool) -> bool:
    """switch the implicit-task-hiddens path of the task-enabled Mlp on / off (tests compare the two paths); returns the old value"""
    global _MLP_HID
    old, _MLP_HID = _MLP_HID, bool(on)
    return old


def mlp_hid_supported(meta1: "LinearMeta", meta2: "LinearMeta", M: int) -> bool:
    key = (M, meta1.K, meta1.N, meta2.N, meta1.r_t, meta2.r_t, meta1.dtype, meta1.mode, meta2.mode, meta1.has_x_tasks, meta2.has_x_tasks,
           _tuning["stream"])
    v = _hid_ok_cache.get(key)
    if v is None:
        d1, d2 = meta1.desc(M), meta2.desc(M)
        v = bool(L.lib().mtlora_mlp_hid_supported(ctypes.byref(d1), ctypes.byref(d2)))
        _hid_ok_cache[key] = v
    return v


def _ensure_packed(meta: "LinearMeta", A_s, B_s, A_t, B_t, device) -> None:
    """the k_hid kernels read the layers' packed factors: a layer without a current FactorPacker buffer packs into a per-call one"""
    if meta.packed is not None:
        return
    buf = torch.empty(packed_bytes(meta), dtype=torch.uint8, device=device)
    d = meta.desc(1)
    d.packed = 0
    st = L.lib().mtlora_linear_pack(ctypes.byref(d), L.ptr(A_s), L.ptr(B_s), L.ptr_array(A_t), L.ptr_array(B_t), L.ptr(buf), buf.numel(),
                                    L.stream_ptr())
    L.check(st, "mtlora_linear_pack")
    meta.packed = buf


class MlpHidFn(torch.autograd.Function):
    """(y_s, y_t[0..T-1]) = fc2(gelu(fc1(x, x_t)), gelu(fc1 task outputs)) of a task-enabled Mlp (swin_transformer_mtlora.py:57-78 of the
    reference with both layers MTLoRALinear called with x_tasks, lora.py:262-266) WITHOUT the 3 T hidden-width task tensors of the
    per-layer path: h_t = h_base + P1_t B1_t^T only enters fc2 through P2_t = s_t gelu(h_t) A2_t^T, and its gradient only feeds
    G, Q1_t and two rank-r_t factor gradients (csrc/hid.hip).  Same parameters, same results up to rounding (the implicit tensors stay
    in fp32 registers instead of being rounded to the compute dtype).

    args: meta1, meta2, x, W1c, W1t, b1, W2c, W2t, b2, A1_s, B1_s, A2_s, B2_s, *x_t(T), *A1_t(T), *B1_t(T), *A2_t(T), *B2_t(T)"""

    NFIX = 13

    @staticmethod
    def forward(ctx, meta1: LinearMeta, meta2: LinearMeta, x, W1c, W1t, b1, W2c, W2t, b2, A1_s, B1_s, A2_s, B2_s, *rest):
        ctx.set_materialize_grads(False)
        T = meta1.T
        x_t = list(rest[:T])
        A1_t, B1_t, A2_t, B2_t = (list(rest[T * k:T * (k + 1)]) for k in range(1, 5))
        L.require_gpu(x, W1c, W2c, *x_t)
        lead = x.shape[:-1]
        x2 = _flat(x, meta1.dtype)
        xt2 = [_flat(t, meta1.dtype) for t in x_t]
        M = x2.numel() // meta1.K
        if x2.shape[-1] != meta1.K or any(t.shape != x2.shape for t in xt2):
            raise RuntimeError(f"mtlora_amd: Mlp input of shape {tuple(x.shape)} for in_features = {meta1.K}")
        dev, dt, H = x.device, meta1.dtype, meta1.N
        f1 = (_f32c(A1_s), _f32c(B1_s), [_f32c(a) for a in A1_t], [_f32c(b) for b in B1_t])
        f2 = (_f32c(A2_s), _f32c(B2_s), [_f32c(a) for a in A2_t], [_f32c(b) for b in B2_t])
        _ensure_packed(meta1, *f1, dev)
        _ensure_packed(meta2, *f2, dev)
        lib = L.lib()
        d1, d2 = meta1.desc(M), meta2.desc(M)
        cb1 = _desc_bytes("ctx", lib.mtlora_linear_ctx_bytes, meta1, M, d1)
        cb2 = _desc_bytes("ctx", lib.mtlora_linear_ctx_bytes, meta2, M, d2)
        if cb1 < 0 or cb2 < 0:
            raise RuntimeError(f"mtlora_amd: invalid Mlp shape M={M} K={meta1.K} hidden={H} N={meta2.N}")
        ctx1 = torch.empty(cb1, dtype=torch.uint8, device=dev)
        ctx2 = torch.empty(cb2, dtype=torch.uint8, device=dev)
        hshape = (*lead, H)
        h_s = torch.empty(hshape, dtype=dt, device=dev)
        a_s = torch.empty(hshape, dtype=dt, device=dev)
        h_base = torch.empty(hshape, dtype=dt, device=dev)
        # fc1: shared output (+ its GELU) and the bare pretrained product; P1 for every segment
        d1.hid, d1.hid_ptr = L.HID_FWD_BASE, h_base.data_ptr()
        st = lib.mtlora_linear_fwd_gelu(ctypes.byref(d1), L.ptr(x2), L.ptr_array(xt2), L.ptr(W1c), L.ptr(b1), L.ptr(f1[0]), L.ptr(f1[1]),
                                        L.ptr_array(f1[2]), L.ptr_array(f1[3]), L.ptr(h_s), L.ptr_array(None), L.ptr(a_s),
                                        L.ptr_array(None), L.ptr(ctx1), cb1, L.stream_ptr())
        L.check(st, "mtlora_linear_fwd_gelu (fc1, implicit task hiddens)")
        d1.hid, d1.hid_ptr = 0, 0
        # the task columns of fc2's P straight from h_base and P1
        key = ("hidfwd", M, meta1.K, H, meta2.N, meta1.r_t, meta2.r_t, dt, _tuning["stream"], _tuning["max_cu"])
        fb = _bytes_cache.get(key)
        if fb is None:
            fb = lib.mtlora_mlp_hid_fwd_scratch_bytes(ctypes.byref(d1), ctypes.byref(d2))
            _bytes_cache[key] = fb
        fscr = torch.empty(fb, dtype=torch.uint8, device=dev)
        st = lib.mtlora_mlp_hid_proj(ctypes.byref(d1), ctypes.byref(d2), L.ptr(h_base), L.ptr(ctx1), L.ptr(ctx2), L.ptr(fscr), fb,
                                     L.stream_ptr())
        L.check(st, "mtlora_mlp_hid_proj")
        oshape = (*lead, meta2.N)
        ys = torch.empty(oshape, dtype=dt, device=dev)
        yt = [torch.empty(oshape, dtype=dt, device=dev) for _ in range(T)]
        d2.hid = L.HID_P_GIVEN
        st = lib.mtlora_linear_fwd(ctypes.byref(d2), L.ptr(a_s), L.ptr_array(None), L.ptr(W2c), L.ptr(b2), L.ptr(f2[0]), L.ptr(f2[1]),
                                   L.ptr_array(f2[2]), L.ptr_array(f2[3]), L.ptr(ys), L.ptr_array(yt), L.ptr(ctx2), cb2, L.stream_ptr())
        L.check(st, "mtlora_linear_fwd (fc2, implicit task hiddens)")
        ctx.meta1, ctx.meta2, ctx.lead, ctx.M = meta1, meta2, lead, M
        ctx.in_dtypes = [x.dtype] + [t.dtype for t in x_t]
        ctx.save_for_backward(x2, W1t, W2t, ctx1, ctx2, h_s, a_s, h_base, *xt2)
        ctx.factor_params = (A1_s, B1_s, *A1_t, *B1_t, A2_s, B2_s, *A2_t, *B2_t)
        return (ys, *yt)

    @staticmethod
    def backward(ctx, *grads):
        meta1, meta2 = ctx.meta1, ctx.meta2
        T, NF = meta1.T, MlpHidFn.NFIX
        n_in = NF + 5 * T
        if all(g is None for g in grads):
            return (None,) * n_in
        x2, W1t, W2t, ctx1, ctx2, h_s, a_s, h_base, *xt2 = ctx.saved_tensors
        M, dev, dt, H = ctx.M, x2.device, meta1.dtype, meta1.N
        g2 = [None if g is None else _flat(g, dt) for g in grads]
        dy_s, dy_t = g2[0], g2[1:1 + T]
        need = ctx.needs_input_grad
        lib = L.lib()
        d1, d2 = meta1.desc(M), meta2.desc(M)
        sb1 = _desc_bytes("bwd", lib.mtlora_linear_bwd_scratch_bytes, meta1, M, d1)
        sb2 = _desc_bytes("bwd", lib.mtlora_linear_bwd_scratch_bytes, meta2, M, d2)
        key = ("hidpart", M, meta1.K, H, meta2.N, meta1.r_t, meta2.r_t, dt, _tuning["stream"], _tuning["max_cu"])
        pb = _bytes_cache.get(key)
        if pb is None:
            pb = lib.mtlora_mlp_hid_bwd_scratch_bytes(ctypes.byref(d1), ctypes.byref(d2))
            _bytes_cache[key] = pb
        scratch1 = torch.empty(sb1, dtype=torch.uint8, device=dev)
        scratch2 = torch.empty(sb2, dtype=torch.uint8, device=dev)
        part = torch.empty(pb, dtype=torch.uint8, device=dev)
        f32 = torch.float32
        hshape, ishape = (*ctx.lead, H), (*ctx.lead, meta1.K)
        dh_s = torch.empty(hshape, dtype=dt, device=dev)
        G = torch.empty(hshape, dtype=dt, device=dev)
        dx = torch.empty(ishape, dtype=dt, device=dev)
        dxt = [torch.empty(ishape, dtype=dt, device=dev) for _ in range(T)]
        # argument positions: A1_s 9, B1_s 10, A2_s 11, B2_s 12, x_t NF.., A1_t NF+T.., B1_t NF+2T.., A2_t NF+3T.., B2_t NF+4T..
        has_s2 = meta2.r_s > 0 and dy_s is not None
        has_s1 = meta1.r_s > 0
        any_t = [dy_t[t] is not None for t in range(T)]
        dA1_s = torch.empty((meta1.r_s, meta1.K), dtype=f32, device=dev) if (has_s1 and need[9]) else None
        dB1_s = torch.empty((H, meta1.r_s), dtype=f32, device=dev) if (has_s1 and need[10]) else None
        dA2_s = torch.empty((meta2.r_s, H), dtype=f32, device=dev) if (has_s2 and need[11]) else None
        dB2_s = torch.empty((meta2.N, meta2.r_s), dtype=f32, device=dev) if (has_s2 and need[12]) else None
        mk = lambda shape, pos, t: torch.empty(shape, dtype=f32, device=dev) if (any_t[t] and need[pos + t]) else None
        dA1_t = [mk((meta1.r_t[t], meta1.K), NF + T, t) for t in range(T)]
        dB1_t = [mk((H, meta1.r_t[t]), NF + 2 * T, t) for t in range(T)]
        dA2_t = [mk((meta2.r_t[t], H), NF + 3 * T, t) for t in range(T)]
        dB2_t = [mk((meta2.N, meta2.r_t[t]), NF + 4 * T, t) for t in range(T)]

        def fc2(phase, stream_ptr):  # dh_s (through gelu'(h_s)), Q2 [phase 1]; dA2_s, dB2_s, dB2_t [phase 2]
            d2.bwd_phase = phase
            st = lib.mtlora_linear_bwd_gelu(ctypes.byref(d2), L.ptr(a_s), L.ptr_array(None), L.ptr(W2t), L.ptr(dy_s), L.ptr_array(dy_t),
                                            L.ptr(ctx2), ctx2.numel(), L.ptr(dh_s), L.ptr_array(None), L.ptr(dA2_s), L.ptr(dB2_s),
                                            L.ptr_array(None), L.ptr_array(dB2_t), L.ptr(scratch2), sb2, L.ptr(h_s), L.ptr_array(None),
                                            stream_ptr)
            L.check(st, "mtlora_linear_bwd_gelu (fc2, implicit task hiddens)")

        def hid(stream_ptr):  # G = dh_s + sum_t dH_t, Q1 task columns, dB1_t, dA2_t
            st = lib.mtlora_mlp_hid_bwd(ctypes.byref(d1), ctypes.byref(d2), L.ptr(h_base), L.ptr(dh_s), L.ptr(ctx1), L.ptr(ctx2),
                                        L.ptr(scratch2), L.ptr(scratch1), L.ptr(G), L.ptr_array(dB1_t), L.ptr_array(dA2_t), L.ptr(part), pb,
                                        stream_ptr)
            L.check(st, "mtlora_mlp_hid_bwd")

        def fc1(phase, stream_ptr):  # dx, dx_t [phase 1]; dA1_s, dB1_s, dA1_t [phase 2]
            d1.bwd_phase = phase
            d1.hid, d1.hid_ptr = L.HID_Q_GIVEN, G.data_ptr()
            st = lib.mtlora_linear_bwd(ctypes.byref(d1), L.ptr(x2), L.ptr_array(xt2), L.ptr(W1t), L.ptr(dh_s), L.ptr_array(None),
                                       L.ptr(ctx1), ctx1.numel(), L.ptr(dx), L.ptr_array(dxt), L.ptr(dA1_s), L.ptr(dB1_s),
                                       L.ptr_array(dA1_t), L.ptr_array(None), L.ptr(scratch1), sb1, stream_ptr)
            L.check(st, "mtlora_linear_bwd (fc1, implicit task hiddens)")

        side = _factor_stream
        fgrads = [t for t in [dA1_s, dB1_s, dA2_s, dB2_s, *dA1_t, *dB1_t, *dA2_t, *dB2_t] if t is not None]
        use_side = (side is not None and fgrads and side.device == dev and M >= _FACTOR_MIN_M and _side_safe_params(ctx.factor_params))
        if _side_join_pending(ctx.factor_params, dev):
            use_side = False
        main = L.stream_ptr()
        if use_side:
            fc2(1, main)
            hid(main)
            fc1(1, main)
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            sp = ctypes.c_void_p(side.cuda_stream)
            fc2(2, sp)
            fc1(2, sp)
            _side_mark_pending(ctx.factor_params, side)
            for t in [x2, ctx1, ctx2, a_s, dh_s, scratch1, scratch2, *xt2, *fgrads] + [g for g in g2 if g is not None]:
                t.record_stream(side)
        else:
            fc2(1, main)
            hid(main)
            fc1(1, main)
            fc2(2, main)
            fc1(2, main)
        dxo = dx if ctx.in_dtypes[0] == dt else dx.to(ctx.in_dtypes[0])
        dxto = [dxt[t] if ctx.in_dtypes[1 + t] == dt else dxt[t].to(ctx.in_dtypes[1 + t]) for t in range(T)]
        return (None, None, dxo, None, None, None, None, None, None, dA1_s, dB1_s, dA2_s, dB2_s, *dxto, *dA1_t, *dB1_t, *dA2_t, *dB2_t)


class GeluDeferredGradFn(torch.autograd.Function):
    """a = gelu(h) (exact erf form) whose backward is the IDENTITY: the consumer is an MTLoRALinear called with
    ``gelu_gate=h``, whose dX kernel multiplies by gelu'(h) itself (``mtlora_linear_bwd_gelu``) -- the standalone GELU
    backward pass (read dA, read h, write dH) disappears.  Only ``Mlp.forward`` pairs the two."""

    @staticmethod
    def forward(ctx, h):
        return torch.nn.functional.gelu(h)

    @staticmethod
    def backward(ctx, g):
        return g


# ----------------------------------------------------------------------------------------------
# window attention core
# ----------------------------------------------------------------------------------------------
@dataclass
class AttnMeta:
    B: int
    H: int
    W: int
    window_size: int
    shift: int
    num_heads: int
    head_dim: int
    image_layout: bool
    scale: float
    mask_value: float = -100.0

    def desc(self, dtype: torch.dtype) -> L.AttnDesc:
        d = L.AttnDesc()
        d.B, d.H, d.W = self.B, self.H, self.W
        d.window_size, d.shift = self.window_size, self.shift
        d.num_heads, d.head_dim = self.num_heads, self.head_dim
        d.image_layout = 1 if self.image_layout else 0
        d.dtype = _DT_CODE[dtype]
        d.scale = self.scale
        d.mask_value = self.mask_value
        return d


class WindowAttentionFn(torch.autograd.Function):
    """out = softmax(scale * q k^T + bias[h] + mask[w]) v per (window, head)  (swin_transformer_mtlora.py:194-220).
    qkv: (..., 3C) [3][nH][hd]; bias: (nH, N, N) fp32 dense.  The shift mask is either ``mask_ids`` (nW, N) int32
    region ids (fast path: mask(i,j) = ids differ ? meta.mask_value : 0) or a general dense ``mask`` (nW, N, N)."""

    @staticmethod
    def forward(ctx, meta: AttnMeta, qkv, bias, mask, mask_ids):
        L.require_gpu(qkv, bias, mask, mask_ids)
        dt = qkv.dtype
        if dt not in _HOT_DTYPES:
            raise RuntimeError(f"mtlora_amd: window attention supports fp32 / bf16 / fp16, got {dt}")
        qkv_c = qkv.contiguous()
        bias_c = bias.detach().float().contiguous()
        if mask_ids is not None:
            mask_ids = mask_ids.to(torch.int32).contiguous()
            mask = None
        elif mask is not None:
            mask = mask.float().contiguous()
        C = meta.num_heads * meta.head_dim
        out = torch.empty(qkv_c.shape[:-1] + (C,), dtype=dt, device=qkv.device)
        d = meta.desc(dt)
        st = L.lib().mtlora_window_attn_fwd(ctypes.byref(d), L.ptr(qkv_c), L.ptr(bias_c), L.ptr(mask), L.ptr(mask_ids),
                                            L.ptr(out), L.stream_ptr())
        L.check(st, "mtlora_window_attn_fwd")
        ctx.meta = meta
        ctx.save_for_backward(qkv_c, bias_c, mask, mask_ids)
        return out

    @staticmethod
    def backward(ctx, dout):
        meta: AttnMeta = ctx.meta
        qkv, bias, mask, mask_ids = ctx.saved_tensors
        dout = dout.to(qkv.dtype).contiguous()
        d = meta.desc(qkv.dtype)
        lib = L.lib()
        sb = lib.mtlora_window_attn_bwd_scratch_bytes(ctypes.byref(d))
        scratch = torch.empty(sb, dtype=torch.uint8, device=qkv.device)
        dqkv = torch.empty_like(qkv)
        dbias = torch.empty_like(bias)
        st = lib.mtlora_window_attn_bwd(ctypes.byref(d), L.ptr(qkv), L.ptr(bias), L.ptr(mask), L.ptr(mask_ids),
                                        L.ptr(dout), L.ptr(dqkv), L.ptr(dbias), L.ptr(scratch), sb, L.stream_ptr())
        L.check(st, "mtlora_window_attn_bwd")
        return None, dqkv, dbias, None, None


# ----------------------------------------------------------------------------------------------
# window process (kernels/window_process/window_process.py)
# ----------------------------------------------------------------------------------------------
def _wp(fn_name: str, src: torch.Tensor, out_shape, B, H, W, C, shift_size, window_size) -> torch.Tensor:
    L.require_gpu(src)
    if not src.is_contiguous():
        raise RuntimeError("input must be contiguous")  # CHECK_CONTIGUOUS, swin_window_process.cpp:65
    if src.numel() != B * H * W * C:
        raise RuntimeError(f"mtlora_amd: {fn_name}: tensor has {src.numel()} elements, expected B*H*W*C={B * H * W * C}")
    out = torch.empty(out_shape, dtype=src.dtype, device=src.device)
    st = getattr(L.lib(), fn_name)(L.ptr(src), L.ptr(out), B, H, W, C, shift_size, window_size,
                                   L.dtype_code(src, allow_f16=True), L.stream_ptr())
    L.check(st, fn_name)
    return out


def roll_and_window_partition_forward(input, B, H, W, C, shift_size, window_size):
    n = B * (H // window_size) * (W // window_size)
    return _wp("mtlora_roll_and_window_partition_forward", input, (n, window_size, window_size, C), B, H, W, C,
               shift_size, window_size)


def roll_and_window_partition_backward(grad_in, B, H, W, C, shift_size, window_size):
    return _wp("mtlora_roll_and_window_partition_backward", grad_in, (B, H, W, C), B, H, W, C, shift_size, window_size)


def window_merge_and_roll_forward(input, B, H, W, C, shift_size, window_size):
    return _wp("mtlora_window_merge_and_roll_forward", input, (B, H, W, C), B, H, W, C, shift_size, window_size)


def window_merge_and_roll_backward(grad_in, B, H, W, C, shift_size, window_size):
    n = B * (H // window_size) * (W // window_size)
    return _wp("mtlora_window_merge_and_roll_backward", grad_in, (n, window_size, window_size, C), B, H, W, C,
               shift_size, window_size)


# ----------------------------------------------------------------------------------------------
# LayerNorm (block glue): reads x once, writes y directly in the dtype the next linear consumes
# ----------------------------------------------------------------------------------------------
def _ln_forward(ctx, x, weight, bias, eps, out_dtype, merge=None):
    """merge=(H, W): x is a (B, H*W, C) token tensor and the normalised rows are its 2x2 neighbourhoods (PatchMerging):
    output (B, H*W/4, 4C), gathered by the kernel."""
    L.require_gpu(x, weight, bias)
    if merge is None:
        C = x.shape[-1]
        x2 = x.reshape(-1, C).contiguous()
        M, mh, mw = x2.shape[0], 0, 0
        out_shape = x.shape
    else:
        mh, mw = merge
        B, Lt, Ct = x.shape
        C, M = 4 * Ct, B * Lt // 4
        x2 = x.contiguous()
        out_shape = (B, Lt // 4, C)
    w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
    y = torch.empty((M, C), dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    st = L.lib().mtlora_layernorm_fwd(L.ptr(x2), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd), M, C,
                                      float(eps), L.dtype_code(x2), L.dtype_code(y), mh, mw, L.stream_ptr())
    L.check(st, "mtlora_layernorm_fwd")
    ctx.save_for_backward(x2, w, mean, rstd)
    ctx.shape, ctx.merge, ctx.MC = x.shape, (mh, mw), (M, C)
    return y.reshape(out_shape)


def _ln_backward(ctx, dy, addend=None):
    """dx (+ addend), dgamma, dbeta"""
    x2, w, mean, rstd = ctx.saved_tensors
    M, C = ctx.MC
    mh, mw = ctx.merge
    dy2 = dy.reshape(M, C).contiguous()
    if dy2.dtype not in _GLUE_DTYPES:
        dy2 = dy2.float()
    add2 = None
    if addend is not None:
        add2 = addend.reshape(x2.shape).to(x2.dtype).contiguous()
    lib = L.lib()
    sb = lib.mtlora_layernorm_bwd_scratch_bytes(M, C, L.dtype_code(x2))
    scratch = torch.empty(sb, dtype=torch.uint8, device=x2.device)
    dx = torch.empty_like(x2)
    dg = torch.empty(C, dtype=torch.float32, device=x2.device)
    db = torch.empty(C, dtype=torch.float32, device=x2.device)
    st = lib.mtlora_layernorm_bwd(L.ptr(dy2), L.ptr(x2), L.ptr(w), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dg),
                                  L.ptr(db), M, C, L.dtype_code(x2), L.dtype_code(dy2), L.ptr(scratch), sb, L.ptr(add2),
                                  mh, mw, L.stream_ptr())
    L.check(st, "mtlora_layernorm_bwd")
    return dx.reshape(ctx.shape), dg, db


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, out_dtype: torch.dtype, merge=None):
        return _ln_forward(ctx, x, weight, bias, eps, out_dtype, merge)

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db = _ln_backward(ctx, dy)
        return dx, dg, db, None, None, None


class LayerNormForkFn(torch.autograd.Function):
    """(x_skip, y) = (x, LayerNorm(x)): the block input feeds the LayerNorm AND the residual connection.  Routing both
    uses through one node lets the backward form  d x = g_skip + LN-backward(g_y)  inside the LayerNorm kernel
    (``dx_addend``) instead of autograd's separate full-size gradient add (one per stream per half block)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, out_dtype: torch.dtype):
        y = _ln_forward(ctx, x, weight, bias, eps, out_dtype)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g_skip, gy):
        if gy is None:  # the normalised output was not used
            return g_skip, None, None, None, None
        dx, dg, db = _ln_backward(ctx, gy, g_skip)
        return dx, dg, db, None, None


class ResidualLayerNormFn(torch.autograd.Function):
    """(x_new, y) = (shortcut + scale[sample] * branch, LayerNorm(x_new)) as ONE kernel; x_new is the next skip path
    (a ``LayerNormForkFn``-style fork), y feeds the following linear.  Backward: d_shortcut = g_x_new +
    LN-backward(g_y), d_branch = scale * d_shortcut, both written by the LayerNorm backward kernel
    (``mtlora_residual_layernorm_fwd/bwd``).  Replaces ResidualDropPathFn + LayerNormForkFn for a single stream."""

    @staticmethod
    def forward(ctx, shortcut, branch, scale, weight, bias, eps: float, out_dtype: torch.dtype):
        L.require_gpu(shortcut, branch, weight, bias)
        C = shortcut.shape[-1]
        B = shortcut.shape[0]
        s2 = shortcut.reshape(-1, C).contiguous()
        b2 = branch.reshape(-1, C).contiguous()
        M = s2.shape[0]
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        x_new = torch.empty_like(s2)
        y = torch.empty((M, C), dtype=out_dtype, device=s2.device)
        mean = torch.empty(M, dtype=torch.float32, device=s2.device)
        rstd = torch.empty(M, dtype=torch.float32, device=s2.device)
        st = L.lib().mtlora_residual_layernorm_fwd(L.ptr(s2), L.ptr(b2), L.ptr(scale), B, L.ptr(w), L.ptr(b), L.ptr(x_new),
                                                   L.ptr(y), L.ptr(mean), L.ptr(rstd), M, C, float(eps), L.dtype_code(s2),
                                                   L.dtype_code(y), L.stream_ptr())
        L.check(st, "mtlora_residual_layernorm_fwd")
        ctx.save_for_backward(x_new, w, mean, rstd, scale)
        ctx.shape, ctx.MC, ctx.B, ctx.bdtype = shortcut.shape, (M, C), B, branch.dtype
        ctx.set_materialize_grads(False)
        return x_new.view(shortcut.shape), y.view(shortcut.shape)

    @staticmethod
    def backward(ctx, g_skip, gy):
        x_new, w, mean, rstd, scale = ctx.saved_tensors
        M, C = ctx.MC
        if gy is None:  # the normalised output was not used: plain residual backward
            if g_skip is None:
                return (None,) * 7
            g = g_skip.reshape(ctx.B, -1, C)
            db_ = g if scale is None else g * scale.view(-1, 1, 1).to(g.dtype)
            return g_skip, db_.to(ctx.bdtype).reshape(ctx.shape), None, None, None, None, None
        dy2 = gy.reshape(M, C).contiguous()
        if dy2.dtype != ctx.bdtype:
            dy2 = dy2.to(ctx.bdtype)
        add2 = None if g_skip is None else g_skip.reshape(M, C).to(x_new.dtype).contiguous()
        lib = L.lib()
        sb = lib.mtlora_layernorm_bwd_scratch_bytes(M, C, L.dtype_code(x_new))
        scratch = torch.empty(sb, dtype=torch.uint8, device=x_new.device)
        dx = torch.empty_like(x_new)
        dbr = torch.empty((M, C), dtype=ctx.bdtype, device=x_new.device)
        dg = torch.empty(C, dtype=torch.float32, device=x_new.device)
        db = torch.empty(C, dtype=torch.float32, device=x_new.device)
        st = lib.mtlora_residual_layernorm_bwd(L.ptr(dy2), L.ptr(x_new), L.ptr(w), L.ptr(mean), L.ptr(rstd), L.ptr(dx),
                                               L.ptr(dbr), L.ptr(dg), L.ptr(db), L.ptr(scale), ctx.B, M, C,
                                               L.dtype_code(x_new), L.dtype_code(dy2), L.ptr(scratch), sb, L.ptr(add2),
                                               L.stream_ptr())
        L.check(st, "mtlora_residual_layernorm_bwd")
        return dx.view(ctx.shape), dbr.view(ctx.shape), None, dg, db, None, None


class ResidualLayerNormMultiFn(torch.autograd.Function):
    """Task-enabled block half: ONE shortcut, n branches -> (x_new_0..n-1, y_0..n-1) with x_new_k = shortcut + scale[k] *
    branch_k and y_k = LayerNorm(x_new_k), one kernel; backward one kernel: d_branch_k = scale[k] * dx_k, d_shortcut = sum_k
    dx_k, dx_k = g_x_new_k + LN'(g_y_k) (never stored), dgamma / dbeta over all streams.
    args: scale ((n, B) fp32 or None), weight, bias, eps, out_dtype, n, shortcut, *branches"""

    @staticmethod
    def forward(ctx, scale, weight, bias, eps: float, out_dtype: torch.dtype, n: int, shortcut, *branches):
        L.require_gpu(shortcut, weight, bias, *branches)
        C, B = shortcut.shape[-1], shortcut.shape[0]
        s2 = shortcut.reshape(-1, C).contiguous()
        bs = [b.reshape(-1, C).contiguous() for b in branches]
        M = s2.shape[0]
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        xs = [torch.empty_like(s2) for _ in range(n)]
        ys = [torch.empty((M, C), dtype=out_dtype, device=s2.device) for _ in range(n)]
        stats = torch.empty((2 * n, M), dtype=torch.float32, device=s2.device)
        means, rstds = [stats[k] for k in range(n)], [stats[n + k] for k in range(n)]
        st = L.lib().mtlora_residual_layernorm_multi_fwd(n, L.ptr(s2), L.ptr_array9(bs), L.ptr(scale), B, L.ptr(w), L.ptr(b),
                                                         L.ptr_array9(xs), L.ptr_array9(ys), L.ptr_array9(means),
                                                         L.ptr_array9(rstds), M, C, float(eps), L.dtype_code(s2),
                                                         L.dtype_code(ys[0]), L.stream_ptr())
        L.check(st, "mtlora_residual_layernorm_multi_fwd")
        ctx.save_for_backward(w, stats, scale, *xs)
        ctx.shape, ctx.MC, ctx.B, ctx.n, ctx.bdtype = shortcut.shape, (M, C), B, n, branches[0].dtype
        return tuple(x.view(shortcut.shape) for x in xs) + tuple(y.view(shortcut.shape) for y in ys)

    @staticmethod
    def backward(ctx, *grads):
        w, stats, scale, *xs = ctx.saved_tensors
        n, (M, C) = ctx.n, ctx.MC
        g_skip, g_y = grads[:n], grads[n:]
        dev = xs[0].device
        dys = [g.reshape(M, C).to(ctx.bdtype).contiguous() for g in g_y]             # (materialised: zeros if unused)
        adds = [g.reshape(M, C).to(xs[0].dtype).contiguous() for g in g_skip]
        lib = L.lib()
        sb = lib.mtlora_layernorm_bwd_scratch_bytes(M, C, L.dtype_code(xs[0]))
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        dsh = torch.empty_like(xs[0])
        dbr = [torch.empty((M, C), dtype=ctx.bdtype, device=dev) for _ in range(n)]
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        means, rstds = [stats[k] for k in range(n)], [stats[n + k] for k in range(n)]
        st = lib.mtlora_residual_layernorm_multi_bwd(n, L.ptr_array9(dys), L.ptr_array9(xs), L.ptr(w), L.ptr_array9(means),
                                                     L.ptr_array9(rstds), L.ptr_array9(adds), L.ptr(dsh), L.ptr_array9(dbr),
                                                     L.ptr(dg), L.ptr(db), L.ptr(scale), ctx.B, M, C, L.dtype_code(xs[0]),
                                                     L.dtype_code(dys[0]), L.ptr(scratch), sb, L.stream_ptr())
        L.check(st, "mtlora_residual_layernorm_multi_bwd")
        return (None, dg, db, None, None, None, dsh.view(ctx.shape), *[d.view(ctx.shape) for d in dbr])


def residual_layer_norm_multi(mod: torch.nn.Module, shortcut: torch.Tensor, branches, drop_prob: float, training: bool):
    """([x_new_k], [mod(x_new_k)]) with x_new_k = shortcut + DropPath_k(branches[k]) (an independent per-sample mask per k):
    the fused multi-stream kernels when they apply, else residual_droppath + layer_norm_fork per stream."""
    C, n = shortcut.shape[-1], len(branches)
    ok = (type(mod) is torch.nn.LayerNorm and mod.elementwise_affine and mod.bias is not None
          and len(mod.normalized_shape) == 1 and shortcut.is_cuda and shortcut.dtype in _GLUE_DTYPES
          and C % 8 == 0 and C <= (1024 if shortcut.dtype == torch.float32 else 1536) and shortcut.dim() == 3
          and 1 <= n <= L.MAX_TASKS + 1 and all(b.shape == shortcut.shape for b in branches) and torch.is_grad_enabled()
          and (shortcut.requires_grad or any(b.requires_grad for b in branches)))
    if ok:
        out_dtype = glue_dtype(shortcut)
        ok = all(b.dtype == out_dtype for b in branches)
    if not ok:
        r = residual_droppath(shortcut, list(branches), drop_prob, training)
        forks = [layer_norm_fork(mod, x) for x in r]
        return [f[0] for f in forks], [f[1] for f in forks]
    scale = None
    if training and drop_prob > 0.0:
        keep = 1.0 - drop_prob
        scale = droppath_scale(n, shortcut.shape[0], keep, shortcut.device)
    outs = ResidualLayerNormMultiFn.apply(scale, mod.weight, mod.bias, mod.eps, out_dtype, n, shortcut, *branches)
    return list(outs[:n]), list(outs[n:])


def residual_layer_norm(mod: torch.nn.Module, shortcut: torch.Tensor, branch: torch.Tensor, drop_prob: float, training: bool):
    """(x_new, mod(x_new)) with x_new = shortcut + DropPath(branch): the fused kernel when it applies (nn.LayerNorm over
    the last dim, branch already in the dtype the LayerNorm output takes), else residual_droppath + layer_norm_fork."""
    C = shortcut.shape[-1]
    ok = (type(mod) is torch.nn.LayerNorm and mod.elementwise_affine and mod.bias is not None
          and len(mod.normalized_shape) == 1 and shortcut.is_cuda and shortcut.dtype in _GLUE_DTYPES
          and C % 8 == 0 and C <= (2048 if shortcut.dtype == torch.float32 else 4096) and branch.shape == shortcut.shape
          and shortcut.dim() == 3 and torch.is_grad_enabled() and (shortcut.requires_grad or branch.requires_grad))
    if ok:
        out_dtype = glue_dtype(shortcut)
        ok = branch.dtype == out_dtype
    if not ok:
        x = residual_droppath(shortcut, [branch], drop_prob, training)[0]
        return layer_norm_fork(mod, x)
    scale = None
    if training and drop_prob > 0.0:
        keep = 1.0 - drop_prob
        scale = droppath_scale(1, shortcut.shape[0], keep, shortcut.device)[0]
    return ResidualLayerNormFn.apply(shortcut, branch, scale, mod.weight, mod.bias, mod.eps, out_dtype)


def layer_norm_fork(mod: torch.nn.Module, x: torch.Tensor):
    """(x for the skip connection, LayerNorm(x) for the following linear): see LayerNormForkFn.  Same dispatch rules as
    ``layer_norm`` (falls back to two separate uses of x when the fused kernel does not apply)."""
    y = layer_norm(mod, x, _fork=True)
    return y if isinstance(y, tuple) else (x, y)


def layer_norm_merge(mod: torch.nn.Module, x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """PatchMerging's ``norm(cat of the 2x2 neighbourhood)`` on a (B, H*W, C) token tensor -> (B, H*W/4, 4C): the gather is
    done by the LayerNorm kernels' addressing (forward reads, backward scatters), not by a strided copy.  Falls back to
    the explicit gather when the kernel does not apply."""
    B, Lt, C = x.shape
    ve = 4 if x.dtype == torch.float32 else 8
    ok = (type(mod) is torch.nn.LayerNorm and mod.elementwise_affine and mod.bias is not None and x.is_cuda
          and x.dtype in _GLUE_DTYPES and C % ve == 0 and Lt == H * W and H % 2 == 0 and W % 2 == 0
          and 4 * C <= (2048 if x.dtype == torch.float32 else 4096))
    if not ok:
        g = x.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, (H // 2) * (W // 2), 4 * C)
        return layer_norm(mod, g)
    return LayerNormFn.apply(x, mod.weight, mod.bias, mod.eps, glue_dtype(x), (H, W))


class LayerNormMergeMultiFn(torch.autograd.Function):
    """PatchMerging's ``norm(2x2 neighbourhood concat)`` applied to n token tensors (B, H*W, C) at once: returns ONE stacked
    (n*B, H*W/4, 4C) tensor (stream-major), so that the following reduction runs as a single GEMM over all streams; one launch
    forward, one backward (+ one reduce) with dgamma / dbeta summed over the streams (``mtlora_layernorm_multi_fwd/bwd``)."""

    @staticmethod
    def forward(ctx, weight, bias, eps: float, out_dtype: torch.dtype, H: int, W: int, n: int, *xs):
        L.require_gpu(weight, bias, *xs)
        B, Lt, Ct = xs[0].shape
        C, M = 4 * Ct, B * Lt // 4
        x2 = [x.contiguous() for x in xs]
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty((n, M, C), dtype=out_dtype, device=x2[0].device)
        stats = torch.empty((2 * n, M), dtype=torch.float32, device=x2[0].device)
        means, rstds = [stats[k] for k in range(n)], [stats[n + k] for k in range(n)]
        st = L.lib().mtlora_layernorm_multi_fwd(n, L.ptr_array9(x2), L.ptr(w), L.ptr(b), L.ptr_array9([y[k] for k in range(n)]),
                                                L.ptr_array9(means), L.ptr_array9(rstds), M, C, float(eps), L.dtype_code(x2[0]),
                                                L.dtype_code(y), H, W, L.stream_ptr())
        L.check(st, "mtlora_layernorm_multi_fwd")
        ctx.save_for_backward(w, stats, *x2)
        ctx.cfg = (n, M, C, H, W, xs[0].shape)
        return y.view(n * B, Lt // 4, C)

    @staticmethod
    def backward(ctx, g):
        w, stats, *x2 = ctx.saved_tensors
        n, M, C, H, W, shape = ctx.cfg
        g3 = g.reshape(n, M, C)
        if g3.dtype not in _GLUE_DTYPES:
            g3 = g3.float()
        g3 = g3.contiguous()
        dev = x2[0].device
        lib = L.lib()
        sb = lib.mtlora_layernorm_multi_bwd_scratch_bytes(n, M, C, L.dtype_code(x2[0]))
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        dxs = [torch.empty_like(x) for x in x2]
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        means, rstds = [stats[k] for k in range(n)], [stats[n + k] for k in range(n)]
        st = lib.mtlora_layernorm_multi_bwd(n, L.ptr_array9([g3[k] for k in range(n)]), L.ptr_array9(x2), L.ptr(w),
                                            L.ptr_array9(means), L.ptr_array9(rstds), L.ptr_array9(dxs), L.ptr(dg), L.ptr(db), M, C,
                                            L.dtype_code(x2[0]), L.dtype_code(g3), L.ptr(scratch), sb, L.ptr_array9(None), H, W,
                                            L.stream_ptr())
        L.check(st, "mtlora_layernorm_multi_bwd")
        return (dg, db, None, None, None, None, None, *[d.view(shape) for d in dxs])


class ResidualMergeNormStreamsFn(torch.autograd.Function):
    """n independent streams: x_new_k = res_k + scale[k] * branch_k (the MLP residual + DropPath of the task-enabled block),
    then PatchMerging's ``norm(2x2 neighbourhood concat)`` of every x_new_k -> ONE stacked (n*B, H*W/4, 4C) tensor.  One launch
    forward, one backward (+ reduce): d_res_k = LN'(g)_k scattered back, d_branch_k = scale[k] * d_res_k; x_new_k is only kept
    for the backward (the block output feeds nothing but the merging).
    args: scale ((n, B) fp32 or None), weight, bias, eps, out_dtype, H, W, n, *res(n), *branches(n)"""

    @staticmethod
    def forward(ctx, scale, weight, bias, eps: float, out_dtype: torch.dtype, H: int, W: int, n: int, *tensors):
        res, brs = tensors[:n], tensors[n:]
        L.require_gpu(weight, bias, *tensors)
        B, Lt, Ct = res[0].shape
        C, M = 4 * Ct, B * Lt // 4
        r2 = [x.contiguous() for x in res]
        b2 = [x.contiguous() for x in brs]
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        xs = [torch.empty_like(x) for x in r2]
        y = torch.empty((n, M, C), dtype=out_dtype, device=r2[0].device)
        stats = torch.empty((2 * n, M), dtype=torch.float32, device=r2[0].device)
        means, rstds = [stats[k] for k in range(n)], [stats[n + k] for k in range(n)]
        st = L.lib().mtlora_residual_layernorm_streams_fwd(n, L.ptr_array9(r2), L.ptr_array9(b2), L.ptr(scale), B, L.ptr(w), L.ptr(b),
                                                           L.ptr_array9(xs), L.ptr_array9([y[k] for k in range(n)]),
                                                           L.ptr_array9(means), L.ptr_array9(rstds), M, C, float(eps),
                                                           L.dtype_code(r2[0]), L.dtype_code(y), H, W, L.stream_ptr())
        L.check(st, "mtlora_residual_layernorm_streams_fwd")
        ctx.save_for_backward(w, stats, scale, *xs)
        ctx.cfg = (n, M, C, H, W, B, res[0].shape, brs[0].dtype)
        return y.view(n * B, Lt // 4, C)

    @staticmethod
    def backward(ctx, g):
        w, stats, scale, *xs = ctx.saved_tensors
        n, M, C, H, W, B, shape, bdt = ctx.cfg
        g3 = g.reshape(n, M, C)
        if g3.dtype != bdt:
            g3 = g3.to(bdt)
        g3 = g3.contiguous()
        dev = xs[0].device
        lib = L.lib()
        sb = lib.mtlora_layernorm_multi_bwd_scratch_bytes(n, M, C, L.dtype_code(xs[0]))
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        dres = [torch.empty_like(x) for x in xs]
        dbr = [torch.empty(shape, dtype=bdt, device=dev) for _ in range(n)]
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        means, rstds = [stats[k] for k in range(n)], [stats[n + k] for k in range(n)]
        st = lib.mtlora_residual_layernorm_streams_bwd(n, L.ptr_array9([g3[k] for k in range(n)]), L.ptr_array9(xs), L.ptr(w),
                                                       L.ptr_array9(means), L.ptr_array9(rstds), L.ptr_array9(dres),
                                                       L.ptr_array9(dbr), L.ptr(dg), L.ptr(db), L.ptr(scale), B, M, C,
                                                       L.dtype_code(xs[0]), L.dtype_code(g3), L.ptr(scratch), sb, L.ptr_array9(None),
                                                       H, W, L.stream_ptr())
        L.check(st, "mtlora_residual_layernorm_streams_bwd")
        return (None, dg, db, None, None, None, None, None, *[d.view(shape) for d in dres], *dbr)


def residual_merge_norm_streams(mod: torch.nn.Module, res, branches, H: int, W: int, drop_prob: float, training: bool):
    """stacked PatchMerging-norm of (res_k + DropPath_k(branches_k)) over the streams, or None when the fused kernels do not
    apply (the caller then forms the residuals and calls ``layer_norm_merge_multi`` / per-stream code)."""
    n = len(res)
    B, Lt, C = res[0].shape
    ve = 4 if res[0].dtype == torch.float32 else 8
    ok = (type(mod) is torch.nn.LayerNorm and mod.elementwise_affine and mod.bias is not None and 2 <= n <= L.MAX_TASKS + 1
          and len(branches) == n and all(x.is_cuda and x.dtype == res[0].dtype and x.shape == res[0].shape for x in res)
          and all(x.shape == res[0].shape and x.dtype == branches[0].dtype for x in branches)
          and res[0].dtype in _GLUE_DTYPES and C % ve == 0 and Lt == H * W and H % 2 == 0 and W % 2 == 0
          and 4 * C <= (2048 if res[0].dtype == torch.float32 else 4096) and torch.is_grad_enabled())
    if ok:
        out_dtype = glue_dtype(res[0])
        ok = branches[0].dtype == out_dtype
    if not ok:
        return None
    scale = None
    if training and drop_prob > 0.0:
        keep = 1.0 - drop_prob
        scale = droppath_scale(n, B, keep, res[0].device)
    return ResidualMergeNormStreamsFn.apply(scale, mod.weight, mod.bias, mod.eps, out_dtype, H, W, n, *res, *branches)


def layer_norm_merge_multi(mod: torch.nn.Module, xs, H: int, W: int):
    """stacked (n*B, H*W/4, 4C) = cat_k PatchMerging-norm(xs[k]) (stream-major), or None when the fused kernel does not apply."""
    B, Lt, C = xs[0].shape
    ve = 4 if xs[0].dtype == torch.float32 else 8
    ok = (type(mod) is torch.nn.LayerNorm and mod.elementwise_affine and mod.bias is not None and 2 <= len(xs) <= L.MAX_TASKS + 1
          and all(x.is_cuda and x.dtype == xs[0].dtype and x.shape == xs[0].shape for x in xs)
          and xs[0].dtype in _GLUE_DTYPES and C % ve == 0 and Lt == H * W and H % 2 == 0 and W % 2 == 0
          and 4 * C <= (2048 if xs[0].dtype == torch.float32 else 4096))
    if not ok:
        return None
    return LayerNormMergeMultiFn.apply(mod.weight, mod.bias, mod.eps, glue_dtype(xs[0]), H, W, len(xs), *xs)


def layer_norm(mod: torch.nn.Module, x: torch.Tensor, feeds_linear: bool = True, _fork: bool = False):
    """``mod(x)`` for an ``nn.LayerNorm`` over the last dim, through the HIP kernel.  ``feeds_linear``: the output is
    only consumed by an MTLoRALinear, so it is written directly in the hot path's compute dtype (bf16 under autocast
    -- no fp32 intermediate + cast pass).  Otherwise (patch_embed.norm, whose output IS the residual stream) the
    output dtype is what the reference produces: fp32 under autocast (autocast runs layer_norm in fp32), else the
    input dtype.  Anything else (custom norm layers, no affine, exotic dtypes) goes to the module itself."""
    ok = (type(mod) is torch.nn.LayerNorm and mod.elementwise_affine and mod.bias is not None
          and len(mod.normalized_shape) == 1 and x.is_cuda and x.dtype in _GLUE_DTYPES
          and x.shape[-1] % 8 == 0 and x.shape[-1] <= (2048 if x.dtype == torch.float32 else 4096))
    if not ok:
        return mod(x)
    if feeds_linear:
        out_dtype = glue_dtype(x)
    else:
        out_dtype = torch.float32 if torch.is_autocast_enabled() else x.dtype
    if _fork and x.requires_grad:
        return LayerNormForkFn.apply(x, mod.weight, mod.bias, mod.eps, out_dtype)  # (x_skip, y)
    return LayerNormFn.apply(x, mod.weight, mod.bias, mod.eps, out_dtype)


# ----------------------------------------------------------------------------------------------
# training-mode BatchNorm (+ReLU) over a channels-last (rows, C) matrix (decoder heads)
# ----------------------------------------------------------------------------------------------
class BatchNormReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum: float, eps: float, relu: bool):
        L.require_gpu(x, weight, bias)
        R, C = x.shape
        x = x.contiguous()
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty_like(x)
        saves = torch.empty(4, C, dtype=torch.float32, device=x.device)  # mean, rstd, scale, shift
        lib = L.lib()
        sb = lib.mtlora_bn_scratch_bytes(R, C, L.dtype_code(x))
        if sb < 0:
            raise RuntimeError(f"mtlora_amd: unsupported BatchNorm shape ({R}, {C}) {x.dtype}")
        scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
        st = lib.mtlora_bn_relu_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(running_mean), L.ptr(running_var), float(momentum),
                                    float(eps), int(relu), L.ptr(y), L.ptr(saves[0]), L.ptr(saves[1]), L.ptr(saves[2]),
                                    L.ptr(saves[3]), R, C, L.dtype_code(x), L.ptr(scratch), sb, L.stream_ptr())
        L.check(st, "mtlora_bn_relu_fwd")
        ctx.save_for_backward(x, saves)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saves = ctx.saved_tensors
        R, C = x.shape
        dy = dy.to(x.dtype).contiguous()
        lib = L.lib()
        sb = lib.mtlora_bn_scratch_bytes(R, C, L.dtype_code(x))
        scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        st = lib.mtlora_bn_relu_bwd(L.ptr(dy), L.ptr(x), L.ptr(saves[0]), L.ptr(saves[1]), L.ptr(saves[2]), L.ptr(saves[3]),
                                    int(ctx.relu), L.ptr(dx), L.ptr(dg), L.ptr(db), R, C, L.dtype_code(x), L.ptr(scratch), sb,
                                    L.stream_ptr())
        L.check(st, "mtlora_bn_relu_bwd")
        return dx, dg, db, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# residual + DropPath over the 1+T tensors of a block half
# ----------------------------------------------------------------------------------------------
class ResidualDropPathFn(torch.autograd.Function):
    """outs[k] = res[k] + scale[k, sample] * ys[k].
    args: scale ((n, B) fp32 or None), shared (bool: every k adds the SAME residual tensor res[0]), n,
          *res (1 tensor if shared else n), *ys (n)."""

    @staticmethod
    def forward(ctx, scale, shared: bool, n: int, *tensors):
        nres = 1 if shared else n
        res, ys = list(tensors[:nres]), list(tensors[nres:nres + n])
        L.require_gpu(*res, *ys)
        B = res[0].shape[0]
        C = res[0].shape[-1]
        M = res[0].numel() // C
        res_c = [r.contiguous() for r in res]
        ys_c = [y.contiguous() for y in ys]
        outs = [torch.empty_like(res_c[0]) for _ in range(n)]
        rlist = res_c * n if shared else res_c
        st = L.lib().mtlora_residual_droppath_fwd(n, L.ptr_array9(rlist), L.ptr_array9(ys_c), L.ptr_array9(outs), L.ptr(scale),
                                                  M, C, B, L.dtype_code(res_c[0]), L.dtype_code(ys_c[0]), L.stream_ptr())
        L.check(st, "mtlora_residual_droppath_fwd")
        ctx.cfg = (shared, n, M, C, B, res_c[0].dtype, ys_c[0].dtype, res_c[0].shape)
        ctx.save_for_backward(scale)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        shared, n, M, C, B, rdt, ydt, shape = ctx.cfg
        (scale,) = ctx.saved_tensors
        if all(g is None for g in grads):
            return (None,) * (3 + (1 if shared else n) + n)
        dev = next(g for g in grads if g is not None).device
        gs = [None if g is None else g.to(rdt).contiguous() for g in grads]
        dys = [None if g is None else torch.empty(shape, dtype=ydt, device=dev) for g in gs]
        dres = torch.empty(shape, dtype=rdt, device=dev) if shared else None
        st = L.lib().mtlora_residual_droppath_bwd(n, L.ptr_array9(gs), L.ptr_array9(dys), L.ptr(dres), L.ptr(scale), M, C, B,
                                                  L.dtype_code(gs[[g is not None for g in gs].index(True)]),
                                                  _DT_CODE.get(ydt, L.F32), L.stream_ptr())
        L.check(st, "mtlora_residual_droppath_bwd")
        dres_out = [dres] if shared else gs          # separate residuals: identity (the incoming gradient itself)
        return (None, None, None, *dres_out, *dys)


def residual_droppath(res, ys, drop_prob: float, training: bool):
    """[res_k + DropPath(ys_k)] for k in range(len(ys)) with an independent per-sample mask per k (timm DropPath,
    scale_by_keep).  ``res``: one tensor shared by all k, or a list of len(ys) tensors."""
    shared = not isinstance(res, (list, tuple))
    rl = [res] if shared else list(res)
    n = len(ys)
    ok = (rl[0].is_cuda and rl[0].shape[-1] % 8 == 0 and all(t.dtype in _GLUE_DTYPES for t in rl + list(ys))
          and len({t.dtype for t in rl}) == 1 and len({t.dtype for t in ys}) == 1 and n <= L.MAX_TASKS + 1
          and all(t.shape == rl[0].shape for t in list(ys) + rl))
    scale = None
    if training and drop_prob > 0.0:
        keep = 1.0 - drop_prob
        scale = droppath_scale(n, rl[0].shape[0], keep, rl[0].device)
    if not ok:
        out = []
        for k in range(n):
            y = ys[k] if scale is None else ys[k] * scale[k].view(-1, *([1] * (ys[k].ndim - 1))).to(ys[k].dtype)
            out.append((rl[0] if shared else rl[k]) + y)
        return out
    return list(ResidualDropPathFn.apply(scale, shared, n, *rl, *ys))


def column_sum(g2: torch.Tensor) -> torch.Tensor:
    """fp32 ``g2.sum(0)`` of a contiguous (M, N) fp32 / bf16 matrix through ``mtlora_colsum`` (deterministic, two launches, no
    device memset: ATen's multi-block reduce zeroes a semaphore buffer first, which a HIP-graph capture of the step cannot
    replay on this stack); shapes the kernel does not take fall back to ATen."""
    M, N = g2.shape
    ve = 4 if g2.dtype == torch.float32 else 8
    if g2.dtype not in (torch.float32, torch.bfloat16) or N % ve or not g2.is_contiguous() or M == 0:
        return g2.sum(0, dtype=torch.float32)
    lib = L.lib()
    sb = lib.mtlora_colsum_scratch_bytes(M, N)
    scratch = torch.empty(sb, dtype=torch.uint8, device=g2.device)
    out = torch.empty(N, dtype=torch.float32, device=g2.device)
    L.check(lib.mtlora_colsum(L.ptr(g2), M, N, L.dtype_code(g2), L.ptr(out), L.ptr(scratch), sb, L.stream_ptr()), "mtlora_colsum")
    return out


def label_stat(lab: torch.Tensor, kind: int, ignore_index: float) -> torch.Tensor:
    """label-only statistic of the fused losses as a 1-element fp32 tensor (kind 0: #{lab != ignore_index}; kind 1:
    mean(1 - (lab >= 0.5))) through ``mtlora_label_stat`` (see ``column_sum`` for why not ATen)."""
    n = lab.numel()
    lib = L.lib()
    sb = lib.mtlora_label_stat_scratch_bytes(n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=lab.device)
    out = torch.empty(1, dtype=torch.float32, device=lab.device)
    L.check(lib.mtlora_label_stat(L.ptr(lab), n, kind, float(ignore_index), L.ptr(out), L.ptr(scratch), sb, L.stream_ptr()),
            "mtlora_label_stat")
    return out


# ----------------------------------------------------------------------------------------------
# loss end of the train step: bilinear upsample (integer scale, align_corners=False) + per-task loss, fused
# ----------------------------------------------------------------------------------------------
LOSS_KINDS = {"softmax": 0, "normals": 1, "balanced_bce": 2}


class UpsampleLossFn(torch.autograd.Function):
    """loss_kind( F.interpolate(low.permute(0,3,1,2), scale_factor=scale, mode="bilinear"), label ) as ONE kernel that also
    produces d loss / d low (csrc/loss.hip); the upsampled prediction never exists.  ``low`` is (B, h, w, C)
    channels-last (fp32 / bf16), ``label`` (B, 1 | C, scale*h, scale*w)."""

    @staticmethod
    def forward(ctx, kind: str, low, label, scale: int, ignore_index: float = 255.0):
        L.require_gpu(low, label)
        B, h, w, C = low.shape
        H, W = h * scale, w * scale
        lab = label.detach().float().contiguous()
        exp_c = C if kind == "normals" else 1
        if tuple(lab.shape) != (B, exp_c, H, W):
            raise RuntimeError(f"mtlora_amd: label shape {tuple(lab.shape)} does not match prediction {(B, exp_c, H, W)}")
        lo = low.detach().contiguous()
        if lo.dtype not in (torch.float32, torch.bfloat16):
            lo = lo.float()
        # label-only statistics (independent of the prediction)
        if kind == "softmax":
            stat = label_stat(lab, 0, ignore_index)
        elif kind == "normals":
            stat = label_stat(lab, 0, ignore_index)
        elif kind == "balanced_bce":
            stat = label_stat(lab, 1, ignore_index)
        else:
            raise RuntimeError(f"mtlora_amd: unknown fused loss kind {kind!r}")
        lib = L.lib()
        n = lib.mtlora_upsample_loss_partials(B, h, w, int(scale))
        if n < 0:
            raise RuntimeError(f"mtlora_amd: fused upsample + loss supports scales 1..32, got {scale}")
        part = torch.empty(max(n, 1), dtype=torch.float32, device=lo.device)
        dlow = torch.empty_like(lo)
        st = lib.mtlora_upsample_loss(LOSS_KINDS[kind], L.ptr(lo), L.ptr(lab), L.ptr(stat), L.ptr(dlow), L.ptr(part), B, h, w,
                                      C, int(scale), L.dtype_code(lo), float(ignore_index), L.stream_ptr())
        L.check(st, "mtlora_upsample_loss")
        ctx.save_for_backward(dlow)
        ctx.in_dtype = low.dtype
        return part[:n].sum() if n > 0 else part.sum() * 0

    @staticmethod
    def backward(ctx, g):
        (dlow,) = ctx.saved_tensors
        return None, (dlow * g.to(dlow.dtype)).to(ctx.in_dtype), None, None, None


# ----------------------------------------------------------------------------------------------
# plain (library) GEMM linears with a huge row count: weight gradient as a batched GEMM over row chunks
# ----------------------------------------------------------------------------------------------
def _layout_weight(weight, bias, cdtype, col_index, n_cols, pad_rows):
    """the weight as the kernel takes it: ``col_index`` scatters its columns into ``n_cols`` (zero elsewhere: the padded channel
    layout of ConcatUpsampleFn), ``pad_rows`` appends zero rows (classes padded to a multiple of 8).  Done inside the Function: no
    autograd nodes of their own for the scatter / cat, the matching gather / slice of the gradient is part of its backward.
    (Round 5 also tried the weight / bias gradients of these linears on the factor-gradient side stream -- nothing in the backward
    chain reads them: c2 -0.7 %, c4 -1.0 %, c5:4 -1.0 %: they leave four concurrent task streams for one.  Removed.)"""
    w = weight.detach().to(cdtype)
    if col_index is not None:
        w = w.new_zeros(w.shape[0], n_cols).index_copy(1, col_index, w)
    b = None if bias is None else bias.detach()
    if pad_rows:
        w = torch.cat([w, w.new_zeros(pad_rows, w.shape[1])], 0)
        b = None if b is None else torch.cat([b, b.new_zeros(pad_rows)], 0)
    return w.contiguous(), b


def _unlayout_grads(dw, db, col_index, pad_rows):
    if pad_rows:
        dw = None if dw is None else dw[:dw.shape[0] - pad_rows]
        db = None if db is None else db[:db.shape[0] - pad_rows].contiguous()
    if col_index is not None and dw is not None:
        dw = dw.index_select(1, col_index)
    elif dw is not None and pad_rows:
        dw = dw.contiguous()
    return dw, db


class SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b on hipBLASLt, like F.linear, but with dW = dY^T X evaluated as a batched GEMM over S row chunks
    + a sum: with M = 100k..400k rows and an output of a few hundred x a few hundred elements the single GEMM runs on
    17-34 workgroups of the 256 CUs (568 us for 1080x270 over 401k rows), the batched form fills the GPU.
    ``weight`` / ``bias`` are the (fp32) master parameters: they are cast to ``cdtype`` inside (no autograd cast nodes) and
    their gradients are returned in the masters' dtype straight from the fp32 chunk sum (no bf16 round trip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, splits: int, zero_bias_grad: bool, cdtype, col_index=None, n_cols=0, pad_rows=0):
        w, bl = _layout_weight(weight, bias, cdtype, col_index, n_cols, pad_rows)
        ctx.save_for_backward(x, w)
        ctx.wlayout = (col_index, pad_rows)
        ctx.has_bias = bias is not None
        ctx.splits = splits
        ctx.zero_bias_grad = zero_bias_grad
        ctx.wdtype = weight.dtype
        ctx.bdtype = None if bias is None else bias.dtype
        return torch.nn.functional.linear(x, w, None if bl is None else bl.to(cdtype))

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        S = ctx.splits
        dx = gy @ w if ctx.needs_input_grad[0] else None

        def wgrads():
            dw = None
            if ctx.needs_input_grad[1]:
                part = torch.bmm(gy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K))  # (S, N, K)
                dw = part.sum(0, dtype=torch.float32).to(ctx.wdtype)
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                # zero_bias_grad: the output feeds a training-mode BatchNorm, whose backward returns columns that sum to zero
                # EXACTLY (dx = scale (dy' - mean(dy') - xhat mean(dy' xhat)), sum(xhat) = 0): the bias gradient is 0, and
                # summing 400k x 1080 elements only to obtain rounding noise costs a full pass over the gradient
                if ctx.zero_bias_grad:
                    db = torch.zeros(N, dtype=ctx.bdtype, device=gy.device)
                else:  # two-stage column sum: a (400k x 21) sum(0) runs on 64 workgroups for 256 us in one stage
                    db = column_sum(gy).to(ctx.bdtype)
            return _unlayout_grads(dw, db, *ctx.wlayout)

        dw, db = wgrads()
        return dx, dw, db, None, None, None, None, None, None


class PlainLinearFn(torch.autograd.Function):
    """y = x W^T + b through the library's fused-GEMM kernel used at rank 0 (``mtlora_linear_fwd/bwd`` with r_s = 0, T = 0):
    for the skinny-K / huge-M GEMMs of the heads (e.g. 100k x 272 -> 1080) hipBLASLt reaches ~1.3 TB/s and ~12 % of the
    MFMA peak, k_nt ~3 TB/s.  dX comes from the same kernel (W^T re-materialised per step: the weight is trained); dW / db
    stay on hipBLASLt as the split-reduction batched GEMM of SplitKLinearFn."""

    @staticmethod
    def forward(ctx, x, weight, bias, splits: int, zero_bias_grad: bool, cdtype, col_index=None, n_cols=0, pad_rows=0):
        L.require_gpu(x, weight)
        M, K = x.shape
        w, bias_l = _layout_weight(weight, bias, cdtype, col_index, n_cols, pad_rows)
        N = w.shape[0]
        meta = LinearMeta(K=K, N=N, r_s=0, r_t=(), scale_s=0.0, scale_t=(), mode=0, has_x_tasks=False, dropout_p=0.0, seed=0,
                          dtype=cdtype)
        d = meta.desc(M)
        lib = L.lib()
        ctx_bytes = _desc_bytes("ctx", lib.mtlora_linear_ctx_bytes, meta, M, d)
        if ctx_bytes < 0:
            raise RuntimeError(f"mtlora_amd: invalid plain-linear shape M={M} K={K} N={N}")
        ctxbuf = torch.empty(max(ctx_bytes, 16), dtype=torch.uint8, device=x.device)
        y = torch.empty((M, N), dtype=cdtype, device=x.device)
        bf = None if bias_l is None else bias_l.float().contiguous()
        st = lib.mtlora_linear_fwd(ctypes.byref(d), L.ptr(x), L.ptr_array(None), L.ptr(w), L.ptr(bf), L.ptr(None), L.ptr(None),
                                   L.ptr_array(None), L.ptr_array(None), L.ptr(y), L.ptr_array(None), L.ptr(ctxbuf), ctx_bytes,
                                   L.stream_ptr())
        L.check(st, "mtlora_linear_fwd (rank 0)")
        ctx.save_for_backward(x, w, ctxbuf)
        ctx.wlayout = (col_index, pad_rows)
        ctx.meta, ctx.splits, ctx.zero_bias_grad = meta, splits, zero_bias_grad
        ctx.has_bias, ctx.wdtype = bias is not None, weight.dtype
        ctx.bdtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, ctxbuf = ctx.saved_tensors
        gy = gy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        S = ctx.splits
        dx = None
        if ctx.needs_input_grad[0]:
            d = ctx.meta.desc(M)
            lib = L.lib()
            sb = _desc_bytes("bwd", lib.mtlora_linear_bwd_scratch_bytes, ctx.meta, M, d)
            scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=x.device)
            wt = w.t().contiguous()
            dx = torch.empty((M, K), dtype=x.dtype, device=x.device)
            st = lib.mtlora_linear_bwd(ctypes.byref(d), L.ptr(x), L.ptr_array(None), L.ptr(wt), L.ptr(gy), L.ptr_array(None),
                                       L.ptr(ctxbuf), ctxbuf.numel(), L.ptr(dx), L.ptr_array(None), L.ptr(None), L.ptr(None),
                                       L.ptr_array(None), L.ptr_array(None), L.ptr(scratch), sb, L.stream_ptr())
            L.check(st, "mtlora_linear_bwd (rank 0)")
        def wgrads():
            dw = None
            if ctx.needs_input_grad[1]:
                if N <= 64:  # narrow output: the library's split-M TN reduction reads x once (hipBLASLt: 0.8 TB/s here)
                    dw = gemm_tn(gy, x).to(ctx.wdtype)
                else:
                    part = torch.bmm(gy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K))
                    dw = part.sum(0, dtype=torch.float32).to(ctx.wdtype)
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if ctx.zero_bias_grad:
                    db = torch.zeros(N, dtype=ctx.bdtype, device=gy.device)
                else:
                    db = column_sum(gy).to(ctx.bdtype)
            return _unlayout_grads(dw, db, *ctx.wlayout)

        dw, db = wgrads()
        return dx, dw, db, None, None, None, None, None, None


def gemm_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(Na, Nb) fp32 = a^T b for row-major (M, Na), (M, Nb) bf16 / fp32 matrices (``mtlora_gemm_tn``)."""
    L.require_gpu(a, b)
    if a.dtype != b.dtype or a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError("gemm_tn: (M, Na) and (M, Nb) matrices of one dtype expected")
    a, b = a.contiguous(), b.contiguous()
    M, Na = a.shape
    Nb = b.shape[1]
    lib = L.lib()
    sb = lib.mtlora_gemm_tn_scratch_bytes(M, Na, Nb)
    if sb < 0:
        raise RuntimeError(f"mtlora_amd: invalid gemm_tn shape M={M} Na={Na} Nb={Nb}")
    scratch = torch.empty(sb, dtype=torch.uint8, device=a.device)
    out = torch.empty((Na, Nb), dtype=torch.float32, device=a.device)
    st = lib.mtlora_gemm_tn(L.ptr(a), L.ptr(b), L.ptr(out), M, Na, Nb, Na, Nb, _DT_CODE[a.dtype],
                            L.ptr(scratch), sb, L.stream_ptr())
    L.check(st, "mtlora_gemm_tn")
    return out


_PLAIN_VIA_KNT = os.environ.get("MTLORA_HEAD_GEMM", "knt") != "blas"


def _big_linear(x, weight, bias, S, feeds_batchnorm, cdtype, col_index=None, pad_rows=0):
    K, N = x.shape[1], weight.shape[0] + pad_rows
    if (_PLAIN_VIA_KNT and cdtype in _HOT_DTYPES and K % 8 == 0 and N % 8 == 0
            and x.dtype == cdtype):
        return PlainLinearFn.apply(x, weight, bias, S, feeds_batchnorm, cdtype, col_index, K, pad_rows)
    return SplitKLinearFn.apply(x, weight, bias, S, feeds_batchnorm, cdtype, col_index, K, pad_rows)


def linear_big_m(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                 feeds_batchnorm: bool = False, col_index: Optional[torch.Tensor] = None, pad_rows: int = 0) -> torch.Tensor:
    """F.linear for (M, K) inputs; switches to SplitKLinearFn when M is large and the weight is trained.
    feeds_batchnorm=True: the output goes straight into a training-mode BatchNorm -> the bias gradient is exactly 0.
    col_index: x has MORE columns than the weight -- column c of the weight multiplies column col_index[c] of x (the padded channel
    layout of ConcatUpsampleFn; the other columns of x meet zeros).  pad_rows: the output gets that many extra all-zero columns."""
    M = x.shape[0]
    if x.dim() == 2 and x.is_cuda and weight.requires_grad and torch.is_grad_enabled() and M >= 1 and x.is_contiguous():
        S = 1  # small M: one chunk -- still this Function, so that the bias gradient is ``column_sum`` (ATen's sum(0) of a mid-size
               # gradient is a multi-block reduce with a device memset, which a HIP-graph capture of the step cannot replay)
        if M >= 16384:
            # enough row chunks that chunks x output tiles (64 x 64) fills the 256 CUs ~8 times (measured on the heads' 1080 x 272
            # gradient: 235 us at 8 chunks, 114 us at 32), each chunk >= 1024 rows
            tiles = ((weight.shape[0] + 63) // 64) * ((weight.shape[1] + 63) // 64)
            S = 4
            while S < 64 and S * tiles < 2048 and M % (2 * S) == 0 and M // (2 * S) >= 1024:
                S *= 2
            if M % S:
                S = 1
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            with torch.autocast("cuda", enabled=False):
                return _big_linear(x.to(dt), weight, bias, S, feeds_batchnorm, dt, col_index, pad_rows)
        return _big_linear(x, weight, bias, S, feeds_batchnorm, x.dtype, col_index, pad_rows)
    if col_index is not None:
        weight = weight.new_zeros(weight.shape[0], x.shape[1]).index_copy(1, col_index, weight)
    if pad_rows:
        weight = torch.cat([weight, weight.new_zeros(pad_rows, weight.shape[1])], 0)
        bias = None if bias is None else torch.cat([bias, bias.new_zeros(pad_rows)], 0)
    return torch.nn.functional.linear(x, weight, bias)


# ----------------------------------------------------------------------------------------------
# HRNet head input: [finest map | upsampled coarse maps] concatenated along channels, without the copies
# ----------------------------------------------------------------------------------------------
def _pad4(c: int) -> int:
    return (c + 3) // 4 * 4


class ConcatUpsampleFn(torch.autograd.Function):
    """maps[i]: (B, h_i, w_i, C_i) channels-last, h_0 = H the finest, H = s_i * h_i.  Returns the (B*H*W, ld) matrix
    [maps[0] | pad | up(maps[1]) | up(maps[2]) | ...] (bilinear, align_corners=False) whose slices start at multiples of
    4 channels (``ConcatUpsampleFn.layout(channels)`` gives offsets and ld): the upsample kernels write / read the slices
    of the concatenated matrix directly (csrc/upsample.hip), replacing F.interpolate x3 + torch.cat and their backward."""

    @staticmethod
    def layout(channels):
        offs, o = [], 0
        for c in channels:
            offs.append(o)
            o += _pad4(c)
        return offs, (o + 7) // 8 * 8

    @staticmethod
    def forward(ctx, *maps):
        L.require_gpu(*maps)
        B, H, W, _ = maps[0].shape
        chans = [m.shape[3] for m in maps]
        offs, ld = ConcatUpsampleFn.layout(chans)
        dt = maps[0].dtype
        out = torch.empty(B * H * W, ld, dtype=dt, device=maps[0].device)
        o3 = out.view(B, H, W, ld)
        o3[..., :chans[0]].copy_(maps[0])
        lib = L.lib()
        end = chans[0]
        for i in range(1, len(maps)):
            if offs[i] > end:
                o3[..., end:offs[i]].zero_()
            m = maps[i].contiguous()
            _, h, w, C = m.shape
            if H % h or W % w or H // h != W // w or C % 4 or m.dtype != dt:
                raise RuntimeError("mtlora_amd: ConcatUpsampleFn needs integer scales, C % 4 == 0 and one dtype")
            st = lib.mtlora_upsample_cl_fwd(L.ptr(m), ctypes.c_void_p(out.data_ptr() + offs[i] * out.element_size()), B, h, w, C,
                                            H // h, ld, L.dtype_code(m), L.stream_ptr())
            L.check(st, "mtlora_upsample_cl_fwd")
            end = offs[i] + C
        if ld > end:
            o3[..., end:].zero_()
        ctx.shapes = [tuple(m.shape) for m in maps]
        ctx.offs, ctx.ld = offs, ld
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, H, W, C0 = ctx.shapes[0]
        ld = ctx.ld
        grads = [g.view(B, H, W, ld)[..., :C0]]
        lib = L.lib()
        for i in range(1, len(ctx.shapes)):
            _, h, w, C = ctx.shapes[i]
            d = torch.empty(ctx.shapes[i], dtype=g.dtype, device=g.device)
            st = lib.mtlora_upsample_cl_bwd(ctypes.c_void_p(g.data_ptr() + ctx.offs[i] * g.element_size()), L.ptr(d), B, h, w, C,
                                            H // h, ld, L.dtype_code(g), L.stream_ptr())
            L.check(st, "mtlora_upsample_cl_bwd")
            grads.append(d)
        return tuple(grads)


# ----------------------------------------------------------------------------------------------
# a run of SwinTransformerBlocks without task outputs: ONE autograd node, one library call per block and direction
# (mtlora_block_fwd / mtlora_block_bwd, ABI v7)
# ----------------------------------------------------------------------------------------------
class BlockCall:
    """what one block of a ``SwinBlockRunFn`` call needs besides its tensors: the four linears' metas (seeds drawn, packed factors
    looked up), their frozen-weight copies, the attention geometry.  Built per call by ``SwinTransformerBlock._block_call``."""
    __slots__ = ("has_norm1", "metas", "weights", "mask", "mask_ids", "H", "W", "num_heads", "window_size", "shift", "hidden",
                 "eps", "attn_scale", "mask_value", "factor_params", "norm_params", "desc", "params", "n_flat")

    def __init__(self, has_norm1, metas, weights, mask, mask_ids, H, W, num_heads, window_size, shift, hidden, eps, attn_scale,
                 factor_params, norm_params=(), mask_value=-100.0):
        self.has_norm1, self.metas, self.weights, self.mask, self.mask_ids = has_norm1, metas, weights, mask, mask_ids
        self.H, self.W, self.num_heads, self.window_size, self.shift, self.hidden = H, W, num_heads, window_size, shift, hidden
        self.eps, self.attn_scale, self.mask_value, self.factor_params = eps, attn_scale, mask_value, factor_params
        self.norm_params = tuple(norm_params)
        self.desc = self.params = None
        self.n_flat = 17 if has_norm1 else 15

    def build_desc(self, B: int, C: int, cdtype: torch.dtype, x_dtype: torch.dtype) -> "L.BlockDesc":
        d = L.BlockDesc()
        d.B, d.H, d.W, d.C, d.hidden = B, self.H, self.W, C, self.hidden
        d.num_heads, d.window_size, d.shift = self.num_heads, self.window_size, self.shift
        d.dtype, d.x_dtype, d.has_norm1 = _DT_CODE[cdtype], _DT_CODE[x_dtype], 1 if self.has_norm1 else 0
        d.eps1, d.eps2, d.eps_next = self.eps
        d.attn_scale, d.mask_value = self.attn_scale, self.mask_value
        M = B * self.H * self.W
        for i, m in enumerate(self.metas):
            d.lin[i] = m.desc(M)
        self.desc = d
        return d


_blk_bytes_cache: dict = {}


def _block_bytes(call: "BlockCall", d, B: int, C: int) -> Tuple[int, int, int]:
    """(save, fwd tmp, bwd scratch) bytes of a block call -- three host calls into the library per new geometry"""
    key = (B, call.H, call.W, C, call.hidden, call.has_norm1, d.dtype, d.x_dtype, call.window_size, call.num_heads, call.shift,
           call.mask_ids is not None, tuple((m.r_s, m.packed is not None) for m in call.metas))
    v = _blk_bytes_cache.get(key)
    if v is None:
        lib = L.lib()
        v = (lib.mtlora_block_save_bytes(ctypes.byref(d)), lib.mtlora_block_fwd_tmp_bytes(ctypes.byref(d)),
             lib.mtlora_block_bwd_scratch_bytes(ctypes.byref(d)))
        if min(v) < 0:
            raise RuntimeError(f"mtlora_amd: invalid SwinTransformerBlock geometry B={B} H={call.H} W={call.W} C={C} hidden={call.hidden}")
        _blk_bytes_cache[key] = v
    return v


class SwinBlockRunFn(torch.autograd.Function):
    """(x_out, normed_out) of a run of n consecutive SwinTransformerBlocks WITHOUT task outputs (reference
    swin_transformer_mtlora.py:326-408, tasks-free path): each block is ONE library call forward and ONE (or phase 1 + phase 2 with the
    factor-gradient side stream) backward; the block calls issue exactly the launches of the per-layer Functions above.

    args: calls (list of BlockCall), x (B, L, C) residual stream, normed (norm1(x) of the first block, or None when that block
    applies it itself), then per block
        bias (nH, N, N) fp32, scale1, scale2 ((B) fp32 DropPath factors or None), [norm1.weight, norm1.bias when has_norm1],
        norm2.weight, norm2.bias, next_norm.weight, next_norm.bias, A_qkv, B_qkv, A_proj, B_proj, A_fc1, B_fc1, A_fc2, B_fc2"""

    @staticmethod
    def forward(ctx, calls, x, normed, *flat):
        L.require_gpu(x, normed)
        lib = L.lib()
        B, _, C = x.shape
        M = x.numel() // C
        cdtype = calls[0].metas[0].dtype
        if not x.is_contiguous() or (normed is not None and (normed.dtype != cdtype or not normed.is_contiguous())):
            raise RuntimeError("mtlora_amd: SwinBlockRunFn needs contiguous inputs, norm1(x) in the compute dtype")
        stream = L.stream_ptr()
        dev = x.device
        tmp = None
        at = 0
        xs, ns, saves = [x], [normed], []
        for c in calls:
            t = flat[at:at + c.n_flat]
            at += c.n_flat
            # every tensor whose raw address goes into BlockParams: on x's device, contiguous, in the dtype the kernels read it as
            # (the per-layer Functions report a mismatch as an error; here it would be an illegal access)
            for q in t:
                if q is not None and (q.device != dev or q.dtype != torch.float32 or not q.is_contiguous()):
                    raise RuntimeError("mtlora_amd: SwinBlockRunFn needs contiguous fp32 bias / DropPath / LayerNorm / factor tensors on the input's device")
            for wc, wt, bf in c.weights:
                if (wc.device != dev or wt.device != dev or wc.dtype != cdtype or wt.dtype != cdtype or not wc.is_contiguous()
                        or not wt.is_contiguous() or (bf is not None and (bf.device != dev or bf.dtype != torch.float32 or not bf.is_contiguous()))):
                    raise RuntimeError("mtlora_amd: SwinBlockRunFn needs the frozen-weight copies in the compute dtype on the input's device")
            if c.mask_ids is not None and (c.mask_ids.device != dev or c.mask_ids.dtype != torch.int32 or not c.mask_ids.is_contiguous()):
                raise RuntimeError("mtlora_amd: SwinBlockRunFn needs a contiguous int32 region-id map on the input's device")
            if c.mask is not None and c.mask_ids is None and (c.mask.device != dev or c.mask.dtype != torch.float32 or not c.mask.is_contiguous()):
                raise RuntimeError("mtlora_amd: SwinBlockRunFn needs a contiguous fp32 attention mask on the input's device")
            d = c.build_desc(B, C, cdtype, x.dtype)
            sb, tb, _ = _block_bytes(c, d, B, C)
            p = L.BlockParams()
            k = 3
            p.attn_bias, p.scale1, p.scale2 = t[0].data_ptr(), (0 if t[1] is None else t[1].data_ptr()), (0 if t[2] is None else t[2].data_ptr())
            if c.has_norm1:
                p.norm1_g, p.norm1_b = t[3].data_ptr(), t[4].data_ptr()
                k = 5
            p.norm2_g, p.norm2_b, p.next_g, p.next_b = t[k].data_ptr(), t[k + 1].data_ptr(), t[k + 2].data_ptr(), t[k + 3].data_ptr()
            for i in range(4):
                wc, wt, bf = c.weights[i]
                p.W[i], p.Wt[i], p.bias[i] = wc.data_ptr(), wt.data_ptr(), (0 if bf is None else bf.data_ptr())
                p.A[i], p.Bf[i] = t[k + 4 + 2 * i].data_ptr(), t[k + 5 + 2 * i].data_ptr()
            p.mask_ids = 0 if c.mask_ids is None else c.mask_ids.data_ptr()
            p.mask = 0 if (c.mask is None or c.mask_ids is not None) else c.mask.data_ptr()
            c.params = p
            if tmp is None or tmp.numel() < tb:
                tmp = torch.empty(tb, dtype=torch.uint8, device=dev)
            save = torch.empty(sb, dtype=torch.uint8, device=dev)
            x_out = torch.empty_like(xs[-1])
            n_out = torch.empty(x.shape, dtype=cdtype, device=dev)
            st = lib.mtlora_block_fwd(ctypes.byref(d), ctypes.byref(p), xs[-1].data_ptr(), 0 if ns[-1] is None else ns[-1].data_ptr(),
                                      x_out.data_ptr(), n_out.data_ptr(), save.data_ptr(), sb, tmp.data_ptr(), tb, stream)
            L.check(st, "mtlora_block_fwd")
            xs.append(x_out)
            ns.append(n_out)
            saves.append(save)
        n = len(calls)
        ctx.calls, ctx.n, ctx.has_normed = calls, n, normed is not None
        ctx.dims = (B, M, C, cdtype)
        # tensors whose addresses the params structs hold: inputs first (autograd tracks their versions), then the call's own buffers
        ctx.save_for_backward(*xs, *[t for t in ns if t is not None], *saves, *[t for t in flat if t is not None])
        ctx.flat_none = [t is None for t in flat]
        return xs[-1], ns[-1]

    @staticmethod
    def backward(ctx, g_x, g_n):
        calls, n = ctx.calls, ctx.n
        B, M, C, cdtype = ctx.dims
        sv = ctx.saved_tensors
        xs = sv[:n + 1]
        k = n + 1
        ns = list(sv[k:k + n + (1 if ctx.has_normed else 0)])
        k += len(ns)
        if not ctx.has_normed:
            ns = [None] + ns
        saves = sv[k:k + n]
        dev = xs[0].device
        lib = L.lib()
        g_x = g_x.contiguous() if g_x.dtype == xs[0].dtype else g_x.to(xs[0].dtype).contiguous()
        g_n = g_n.contiguous() if g_n.dtype == cdtype else g_n.to(cdtype).contiguous()
        side = _factor_stream
        grads_flat = []
        for bi in range(n - 1, -1, -1):
            c = calls[bi]
            d, p = c.desc, c.params
            _, _, scb = _block_bytes(c, d, B, C)
            scratch = torch.empty(scb, dtype=torch.uint8, device=dev)
            nH, N = c.num_heads, c.window_size * c.window_size
            gx = torch.empty_like(xs[bi])
            gn = None if c.has_norm1 else torch.empty(xs[bi].shape, dtype=cdtype, device=dev)
            lnw = torch.empty((6, C), dtype=torch.float32, device=dev)
            dbias = torch.empty((nH, N, N), dtype=torch.float32, device=dev)
            fg = []
            for i, m in enumerate(c.metas):
                fg.append(torch.empty((m.r_s, m.K), dtype=torch.float32, device=dev))
                fg.append(torch.empty((m.N, m.r_s), dtype=torch.float32, device=dev))
            g = L.BlockGrads()
            g.g_x, g.g_normed = gx.data_ptr(), (0 if gn is None else gn.data_ptr())
            g.d_norm1_g, g.d_norm1_b, g.d_norm2_g, g.d_norm2_b = lnw[0].data_ptr(), lnw[1].data_ptr(), lnw[2].data_ptr(), lnw[3].data_ptr()
            g.d_next_g, g.d_next_b, g.dbias = lnw[4].data_ptr(), lnw[5].data_ptr(), dbias.data_ptr()
            for i in range(4):
                g.dA[i], g.dB[i] = fg[2 * i].data_ptr(), fg[2 * i + 1].data_ptr()
            # everything phase 2 writes on the side stream: the eight factor gradients and the LayerNorm dgamma / dbeta (their
            # second-stage reduces ride along, csrc/internal.h) -- none of their Parameters may have a .grad to accumulate into
            fparams = c.factor_params + c.norm_params
            use_side = side is not None and side.device == dev and M >= _FACTOR_MIN_M and _side_safe_params(fparams)
            if _side_join_pending(fparams, dev):
                use_side = False
            args = (ctypes.byref(d), ctypes.byref(p), xs[bi].data_ptr(), 0 if ns[bi] is None else ns[bi].data_ptr(),
                    xs[bi + 1].data_ptr(), g_x.data_ptr(), g_n.data_ptr(), saves[bi].data_ptr(), saves[bi].numel(), ctypes.byref(g),
                    scratch.data_ptr(), scb)
            if use_side:
                L.check(lib.mtlora_block_bwd(*args, 1, L.stream_ptr()), "mtlora_block_bwd")
                ev = torch.cuda.Event()
                ev.record()
                side.wait_event(ev)
                L.check(lib.mtlora_block_bwd(*args, 2, ctypes.c_void_p(side.cuda_stream)), "mtlora_block_bwd (factor gradients)")
                _side_mark_pending(fparams, side)
                for t in (xs[bi], xs[bi + 1], saves[bi], scratch, lnw, g_n, *fg) + (() if ns[bi] is None else (ns[bi],)):
                    t.record_stream(side)  # allocated on this stream, still in use on the side stream when freed here
            else:
                L.check(lib.mtlora_block_bwd(*args, 0, L.stream_ptr()), "mtlora_block_bwd")
            blk = [dbias, None, None]
            if c.has_norm1:
                blk += [lnw[0], lnw[1]]
            blk += [lnw[2], lnw[3], lnw[4], lnw[5], *fg]
            grads_flat = blk + grads_flat
            g_x, g_n = gx, gn
        return (None, g_x, g_n if ctx.has_normed else None, *grads_flat)
