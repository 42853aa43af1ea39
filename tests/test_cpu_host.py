"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares, host logic of the
drop-in modules (construction, names, trainable set, failure modes), and the 2-rank gradient reducer on gloo."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from mtlora_amd.csrc.build import build
    return build(verbose=False)


def test_header_symbols_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "mtlora_hip.h")).read()
    declared = set(re.findall(r"\b(mtlora_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mtlora_linear_desc", "mtlora_attn_desc", "mtlora_prof_summary"}
    from mtlora_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()  # loads, binds every symbol (AttributeError if one is missing), checks the ABI version
    assert L.mtlora_version() == _lib.ABI_VERSION
    assert L.mtlora_error_string(-5).decode().startswith("ctx/scratch")
    nm = subprocess.run(["nm", "-D", built_lib], capture_output=True, text=True).stdout
    for s in declared:
        assert f" T {s}" in nm, s


def test_header_prototypes_match_ctypes_signatures():
    """every prototype of include/mtlora_hip.h has as many parameters as the ctypes signature that binds it, and the
    two descriptor structs have the sizes the header implies (ABI drift check, no GPU needed)."""
    import ctypes
    from mtlora_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mtlora_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = dict(re.findall(r"\b(mtlora_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr))
    for name, (_, argtypes) in _lib._SIGS.items():
        assert name in protos, name
        params = protos[name].strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(argtypes), (name, n, len(argtypes))
    # struct layouts as the C compiler sees them (gcc on the header) == the ctypes mirrors
    import tempfile
    src = '''#include <stdio.h>
#include <stddef.h>
#include "mtlora_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mtlora_linear_desc), offsetof(mtlora_linear_desc, scale_t),
           offsetof(mtlora_linear_desc, seed), offsetof(mtlora_linear_desc, seed_offset), sizeof(mtlora_attn_desc),
           offsetof(mtlora_attn_desc, mask_value));
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        c, exe = os.path.join(td, "l.c"), os.path.join(td, "l")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    LD, AD = _lib.LinearDesc, _lib.AttnDesc
    assert got == [ctypes.sizeof(LD), LD.scale_t.offset, LD.seed.offset, LD.seed_offset.offset, ctypes.sizeof(AD),
                   AD.mask_value.offset], got


def test_size_queries_and_validation(built_lib):
    """pure host calls (no GPU): shape validation and workspace sizing of the C ABI."""
    import ctypes
    from mtlora_amd import _lib
    L = _lib.lib()
    d = _lib.LinearDesc()
    d.M, d.K, d.N, d.dtype, d.T, d.r_s = 1000, 96, 384, _lib.BF16, 4, 64
    for i in range(4):
        d.r_t[i] = 4
    n = L.mtlora_linear_ctx_bytes(ctypes.byref(d))
    assert n >= 1000 * 128 * 2  # P: M x (64 + 4*16) bf16
    assert L.mtlora_linear_bwd_scratch_bytes(ctypes.byref(d)) > 0
    d.K = 100  # not a multiple of 8
    assert L.mtlora_linear_ctx_bytes(ctypes.byref(d)) < 0
    assert L.mtlora_linear_fwd(ctypes.byref(d), None, None, None, None, None, None, None, None, None, None, None, 0, None) == -3
    a = _lib.AttnDesc()
    a.B, a.H, a.W, a.window_size, a.shift, a.num_heads, a.head_dim, a.dtype = 2, 14, 14, 7, 3, 3, 32, _lib.BF16
    assert L.mtlora_window_attn_bwd_scratch_bytes(ctypes.byref(a)) > 0
    a.head_dim = 64
    assert L.mtlora_window_attn_bwd_scratch_bytes(ctypes.byref(a)) < 0  # unsupported head_dim
    # null pointers / bad shapes are rejected before any launch
    assert L.mtlora_roll_and_window_partition_forward(None, None, 1, 14, 14, 8, -3, 7, 0, None) == -4
    assert L.mtlora_roll_and_window_partition_forward(ctypes.c_void_p(16), ctypes.c_void_p(32), 1, 14, 14, 8, -3, 5, 0, None) == -2


def test_descriptor_selection_fields_are_validated(built_lib):
    """ABI v6: kernel selection travels in the descriptor (no environment reads in the library); out-of-range values are rejected
    by the size queries like any other invalid field, and the Python side fills them from functional.set_tuning."""
    import ctypes
    from mtlora_amd import _lib
    from mtlora_amd import functional as Fn
    L = _lib.lib()
    d = _lib.LinearDesc()
    d.M, d.K, d.N, d.dtype, d.T, d.r_s = 1000, 96, 384, _lib.BF16, 0, 64
    base = L.mtlora_linear_ctx_bytes(ctypes.byref(d))
    assert base > 0
    for field, bad in (("sel_stream", 2), ("sel_dense", 5), ("sel_tn", -1), ("sel_projk", 7), ("max_cu", -1)):
        setattr(d, field, bad)
        assert L.mtlora_linear_ctx_bytes(ctypes.byref(d)) < 0, field
        setattr(d, field, 0)
    d.sel_dense, d.sel_tn, d.sel_projk, d.max_cu = 2, 2, 2, 1
    assert L.mtlora_linear_ctx_bytes(ctypes.byref(d)) == base  # workspace sizes do not depend on the selection
    prev = Fn.set_tuning(stream=1, dense=2, max_cu=3)
    try:
        meta = Fn.LinearMeta(K=96, N=96, r_s=8, r_t=(), scale_s=1.0, scale_t=(), mode=0, has_x_tasks=False, dropout_p=0.0, seed=0,
                             dtype=torch.bfloat16)
        dd = meta.desc(64)
        assert (dd.sel_stream, dd.sel_dense, dd.sel_tn, dd.sel_projk, dd.max_cu) == (1, 2, prev["tn"], prev["projk"], 3)
        with pytest.raises(KeyError):
            Fn.set_tuning(nonsense=1)
    finally:
        Fn.set_tuning(**prev)
    # the shipped library does not read the environment for kernel selection / debugging
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"MTLORA_SP\0", b"MTLORA_NTD", b"MTLORA_SP_TN", b"MTLORA_SP_PROJK", b"MTLORA_NT_DBG", b"MTLORA_SP_DBG", b"MTLORA_SP_STG",
                 b"MTLORA_PS_MAXT", b"MTLORA_RANK_OUT"):
        assert name not in blob, name


def test_partial_state_dict_into_merged_layer_unmerges_first():
    """loading a LoRA-only (strict=False) dict into a MERGED layer must not leave W + s B_old A_old behind with the
    merged flag cleared (the shared update would be added twice; a later unmerge would subtract the wrong delta)."""
    from mtlora_amd.lora import MTLoRALinear
    torch.manual_seed(0)
    m = MTLoRALinear(16, 24, r=4, lora_shared_scale=2.0)
    with torch.no_grad():
        m.lora_shared_B.normal_()
    w0 = m.linear.weight.detach().clone()
    m.eval()
    assert m.merge() and m.merged
    assert not torch.allclose(m.linear.weight, w0)
    new = {"lora_shared_A": torch.randn(4, 16), "lora_shared_B": torch.randn(24, 4)}
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert "linear.weight" in missing and not unexpected
    assert not m.merged
    assert torch.allclose(m.linear.weight, w0, atol=1e-6)      # the OLD delta was taken out before the factors changed
    assert torch.equal(m.lora_shared_A, new["lora_shared_A"])
    # a full dict into a merged layer: the loaded weight is un-merged by definition
    m.merge()
    full = {k: v.clone() for k, v in m.state_dict().items()}
    assert torch.allclose(full["linear.weight"], w0, atol=1e-6)  # state_dict of a merged layer saves the un-merged weight
    m.load_state_dict(full)
    assert not m.merged and torch.allclose(m.linear.weight, w0, atol=1e-6)


def test_no_cpu_fallback():
    from mtlora_amd.lora import MTLoRALinear
    from mtlora_amd import window_process as WP
    m = MTLoRALinear(96, 96, r=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 96))
    with pytest.raises(RuntimeError):
        WP.WindowProcess.apply(torch.randn(1, 7, 7, 8), 1, 7, 7, 8, 0, 7)


def test_module_api_parity(golden):
    """constructor semantics of reference lora.py:161-247 and the state dict / trainable set of the C2 model."""
    from mtlora_amd.lora import LoRALayer, MTLoRALinear, mark_only_lora_as_trainable
    from mtlora_amd import mtl_harness as H
    m = MTLoRALinear(96, 288, r=8, lora_shared_scale=4.0, lora_dropout=0.05, bias=False)
    assert isinstance(m, LoRALayer) and m.r == 8 and m.tasks is None and m.shared_mode == "matrix" and not m.merged
    assert m.linear.bias is None and isinstance(m.lora_dropout, torch.nn.Dropout)
    assert m.lora_shared_B.abs().sum() == 0 and m.lora_shared_A.abs().sum() > 0   # B = 0, A ~ kaiming
    m = MTLoRALinear(96, 96, r={"shared": 8, "a": 4}, lora_task_scale={"a": 2.0}, tasks=["a"], shared_mode="add")
    assert m.shared_mode == "addition" and hasattr(m, "lora_norm") and not hasattr(m, "lora_shared_A")
    m = MTLoRALinear(96, 96, r={"shared": 8, "a": 4}, lora_task_scale={"a": 2.0}, tasks=["a"], shared_mode="lora_only")
    assert m.tasks is None and m.shared_mode == "matrix"
    m = MTLoRALinear(96, 96, r=0)
    assert not hasattr(m, "lora_shared_A")
    assert m.merge() is False and not m.merged          # nothing to merge at rank 0 (the reference's merge() is a stub)
    # merge(): W' = W + s B A for layers whose every output sees the shared update; 'matrix' + tasks stays unmerged
    m = MTLoRALinear(16, 24, r=4, lora_shared_scale=2.0)
    torch.nn.init.normal_(m.lora_shared_B, std=0.1)
    w0 = m.linear.weight.detach().clone()
    with pytest.raises(RuntimeError):                   # merging is an inference-time operation: refused in train mode
        m.merge()
    m.eval()
    assert m.merge() and m.merged and not m.merge()
    assert torch.allclose(m.linear.weight, w0 + 2.0 * m.lora_shared_B @ m.lora_shared_A, atol=1e-6)
    # a merged layer SAVES the un-merged weight (W + s B A next to A and B would apply the update twice after a reload) ...
    sd = m.state_dict()
    assert torch.allclose(sd["linear.weight"], w0, atol=1e-6) and m.merged
    fresh = MTLoRALinear(16, 24, r=4, lora_shared_scale=2.0)
    fresh.load_state_dict(sd)
    assert not fresh.merged and torch.allclose(fresh.linear.weight, w0, atol=1e-6)
    # ... and loading a checkpoint INTO a merged layer resets the flag (the loaded weight never contained the delta)
    m.load_state_dict(sd)
    assert not m.merged and torch.allclose(m.linear.weight, w0, atol=1e-6)
    assert m.merge() and m.merged
    m.train()                                            # train() un-merges
    assert not m.merged and torch.allclose(m.linear.weight, w0, atol=1e-6)
    mt = MTLoRALinear(16, 24, r={"shared": 4, "a": 2}, lora_task_scale={"a": 2.0}, tasks=["a"]).eval()
    assert mt.merge() is False
    mv2 = MTLoRALinear(16, 24, r={"shared": 4, "a": 2}, lora_task_scale={"a": 2.0}, tasks=["a"], shared_mode="matrixv2").eval()
    assert mv2.merge() is True and mv2.unmerge() is True
    c = golden("c2_structure.pt")
    model = H.build_model(img_size=448, freeze=True)
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == c["state"]
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == sorted(c["trainable"])
    assert sum(p.numel() for p in model.parameters()) == c["n_params"] == 34262906
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == c["n_trainable"] == 8344634
    # mark_only_lora_as_trainable flags (reference lora.py:580-630)
    model = H.build_model(img_size=224, tasks=("semseg",), freeze=False)
    mark_only_lora_as_trainable(model.backbone, bias="none", freeze_patch_embed=True, freeze_norm=True,
                                free_relative_bias=True, freeze_downsample_reduction=True)
    assert all("lora_" in n for n, p in model.backbone.named_parameters() if p.requires_grad)
    mark_only_lora_as_trainable(model.backbone, bias="all")
    assert any(n.endswith("linear.bias") and p.requires_grad for n, p in model.backbone.named_parameters())


def test_checkpoint_key_mapping():
    """vanilla-Swin keys -> MTLoRA keys (reference lora.py:644-668, utils.py:125-149)."""
    from mtlora_amd.lora import map_old_state_dict_weights
    sd = {"layers.0.blocks.0.attn.qkv.weight": torch.arange(12.).reshape(6, 2), "layers.0.blocks.0.attn.qkv.bias": torch.arange(6.)}
    mapping = {"attn.qkv.weight": "attn.qkv.linear.weight", "attn.qkv.bias": "attn.qkv.linear.bias", "x.y": "x.z"}
    out = map_old_state_dict_weights(dict(sd), mapping, "layers.0.blocks.0.")
    assert set(out) == {"layers.0.blocks.0.attn.qkv.linear.weight", "layers.0.blocks.0.attn.qkv.linear.bias"}
    out = map_old_state_dict_weights(dict(sd), mapping, "layers.0.blocks.0.", split_qkv=True)
    assert torch.equal(out["layers.0.blocks.0.attn.qkv.k.linear.weight"], torch.arange(12.).reshape(6, 2)[2:4])


def test_harness_losses_match_golden(golden):
    from mtlora_amd import mtl_harness as H
    for t, d in golden("losses.pt").items():
        pred = d["pred"].clone().requires_grad_(True)
        l = H.task_loss(t, pred, d["label"])
        assert abs(l.item() - d["loss"]) < 1e-5 * max(1.0, abs(d["loss"])), t
        l.backward()
        assert torch.allclose(pred.grad, d["dpred"], rtol=1e-4, atol=1e-7), t


def test_load_checkpoint_matches_reference(golden, tmp_path):
    """mtlora_amd.checkpoint.load_checkpoint == reference utils.load_checkpoint (utils.py:41-176) on a synthetic vanilla
    Swin checkpoint loaded into an MTLoRA backbone (fixture captured from the real reference, make_golden.gen_checkpoint):
    .weight -> .linear.weight mapping, attn_mask strip, relative-position table bicubic 13x13 -> 7x7, the set of tensors
    that change, the loaded values and the missing / unexpected key report."""
    import logging
    from mtlora_amd import checkpoint as C
    from mtlora_amd import mtl_harness as H
    from mtlora_amd.swin_transformer_mtlora import SwinTransformerMTLoRA
    from oracle import mtlora_oracle as O
    c = golden("checkpoint_map.pt")
    sd = {}
    for k, (shape, dt) in c["ckpt"].items():
        if dt.startswith("torch.float") and "attn_mask" not in k:
            sd[k] = O.det_tensor("ckpt." + k, shape, 0.05)
        else:
            sd[k] = torch.zeros(shape, dtype=getattr(torch, dt.split(".")[1]))
    tasks = ["semseg", "normals"]
    mt = H.mtlora_namespace(tasks, r_shared=8, r_task=4, n_stages=2, SPLIT_QKV=False)
    tgt = SwinTransformerMTLoRA(img_size=64, patch_size=4, in_chans=3, num_classes=0, embed_dim=48, depths=[2, 2],
                                num_heads=[2, 4], window_size=4, drop_path_rate=0.0, tasks=tasks, mtlora=mt)
    O.det_fill_(tgt.named_parameters())
    before = {k: v.clone() for k, v in tgt.state_dict().items()}
    path = tmp_path / "vanilla.pth"
    torch.save({"model": sd}, path)

    class Log(logging.Logger):
        def __init__(self):
            super().__init__("t")
            self.w = []

        def info(self, m, *a, **k):
            pass

        def warning(self, m, *a, **k):
            self.w.append(str(m))

    log = Log()
    cfg = H.AttrDict(MODEL=H.AttrDict(RESUME="", RESUME_BACKBONE=str(path), MTLORA=mt, UPDATE_RELATIVE_POSITION=True),
                     TRAIN=H.AttrDict(SKIP_DECODER_CKPT=False), EVAL_MODE=True)
    assert C.load_checkpoint(cfg, tgt, None, None, None, log, backbone=True) == 0.0
    after = tgt.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    assert changed == c["changed"]
    assert log.w == c["warnings"]
    for k, cs in c["loaded"].items():
        f = after[k].double().flatten()
        assert tuple(after[k].shape) == tuple(cs["shape"]), k
        assert abs(f.sum().item() - cs["sum"]) <= 1e-6 * max(1.0, cs["abssum"]), k
        assert torch.allclose(f[cs["idx"]], cs["samples"], rtol=1e-6, atol=1e-8), k
    for k, t in c["tables"].items():
        assert torch.allclose(after[k], t, rtol=1e-6, atol=1e-7), k


def test_dropout_mask_statistics():
    from oracle import mtlora_oracle as O
    for p in (0.05, 0.25, 0.5):
        k = O.dropout_keep_mask(0xABCDEF0123456789, 0, 2000, 96, p)
        assert abs(k.float().mean().item() - (1 - p)) < 0.01
        assert abs(k.float().mean(0).std().item()) < 0.03  # no column bias
    a = O.dropout_keep_mask(1, 0, 64, 64, 0.5)
    b = O.dropout_keep_mask(2, 0, 64, 64, 0.5)
    assert (a != b).float().mean() > 0.3


_DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mtlora_amd.ddp import GradReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://")
torch.manual_seed(rank)                        # replicas start DIFFERENT: the reducer must broadcast rank 0's parameters / buffers
net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
buf = torch.full((5,), float(rank))
net[0].weight.requires_grad_(False)            # frozen parameters never enter a bucket
unused = torch.nn.Parameter(torch.zeros(3))     # trainable but never used -> grad stays None
class _Half(torch.autograd.Function):          # like MTLoRALinearFn for an unused output: returns an UNDEFINED gradient for
    @staticmethod                              # `b`; torch still runs b's accumulation node and its post-accumulate hook
    def forward(ctx, x, a, b):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x)
        return x * a
    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g, (g * x).sum(0), None
ha, hb = torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(4))
_fwd = net.forward
net.forward = lambda x: _Half.apply(_fwd(x), ha, hb)
params = list(net.parameters()) + [unused, ha, hb]
red = GradReducer(params, bucket_mb=0.0002, buffers=[buf])     # several buckets
chk = torch.cat([p.detach().reshape(-1) for p in params] + [buf])
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
assert all(torch.equal(both[0], v) for v in both) and buf.eq(0).all()   # frozen weight, trainables and buffers == rank 0's
assert red.nbytes == 4 * sum(p.numel() for p in params if p.requires_grad)
g = torch.Generator().manual_seed(100)
xs = torch.randn(world, 5, 8, generator=g)
red.prepare(); net(xs[rank]).pow(2).sum().backward(); red.finish()
# single-process reference on the concatenated batch: mean over ranks of per-rank grads
ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
ref.load_state_dict(net.state_dict())
sum(ref(xs[r]).pow(2).sum() for r in range(world)).backward()
for (n, p), q in zip(net.named_parameters(), ref.parameters()):
    if p.requires_grad:
        assert torch.allclose(p.grad, q.grad / world, rtol=1e-5, atol=1e-6), n
    else:
        assert p.grad is None
assert unused.grad is None and hb.grad is None and ha.grad is not None
# second step reuses the buckets
net.zero_grad(); red.prepare(); net(xs[rank] * 2).pow(2).sum().backward(); red.finish()
t = torch.stack([p.grad.sum() for p in net.parameters() if p.requires_grad]).sum().reshape(1)
gathered = [torch.zeros(1) for _ in range(world)]
dist.all_gather(gathered, t)
assert all(torch.allclose(gathered[0], v) for v in gathered)   # every rank holds the same averaged gradients
# a parameter that only rank 1 uses: rank 0 must still receive the mean (its own share is zero), not keep grad None
lone = torch.nn.Parameter(torch.ones(4))
w = torch.nn.Parameter(torch.ones(4))
red2 = GradReducer([w, lone], bucket_mb=1.0)
for step in range(2):
    w.grad = lone.grad = None
    red2.prepare()
    y = (w * (rank + 1.0)).sum() + ((lone * 3.0).sum() if rank == 1 else 0.0)
    y.backward(); red2.finish()
    assert torch.allclose(w.grad, torch.full((4,), 1.5)), w.grad
    assert lone.grad is not None and torch.allclose(lone.grad, torch.full((4,), 1.5)), (rank, lone.grad)
dist.destroy_process_group()
print("OK", rank)
'''


def test_grad_reducer_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_DDP_WORKER)
    port = 29500 + (os.getpid() % 500)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_droppath_pool_plan_and_fallback():
    """functional._DropPathPool (host logic, device-agnostic): the requests of one announced step are the plan of the next; a
    planned step serves views of ONE bulk tensor with values in {0, 1/keep}; a deviating request falls back to its own draw and
    re-records the plan; outside an announced step every request draws on its own."""
    from mtlora_amd import functional as Fn
    Fn.droppath_reset()
    d = torch.device("cpu")
    reqs = [(1, 16, 0.9), (5, 16, 0.8), (1, 16, 0.5)]
    lone = Fn.droppath_scale(2, 16, 0.7, d)  # no step announced
    assert lone.shape == (2, 16)
    Fn.droppath_begin_step(d)
    a = [Fn.droppath_scale(n, B, k, d) for n, B, k in reqs]
    Fn.droppath_end_step()
    assert len({t.untyped_storage().data_ptr() for t in a}) == 3  # recorded, drawn one by one
    torch.manual_seed(1)
    Fn.droppath_begin_step(d)
    b = [Fn.droppath_scale(n, B, k, d) for n, B, k in reqs]
    Fn.droppath_end_step()
    assert len({t.untyped_storage().data_ptr() for t in b}) == 1  # one bulk draw
    for (n, B, k), t in zip(reqs, b):
        assert t.shape == (n, B) and bool(((t == 0) | ((t - 1.0 / k).abs() < 1e-6)).all())
    torch.manual_seed(1)
    Fn.droppath_begin_step(d)
    c = [Fn.droppath_scale(n, B, k, d) for n, B, k in reqs]
    Fn.droppath_end_step()
    assert all(torch.equal(x, y) for x, y in zip(b, c))  # reproducible under the global generator
    Fn.droppath_begin_step(d)
    first = Fn.droppath_scale(1, 16, 0.9, d)          # as planned
    odd = Fn.droppath_scale(3, 16, 0.8, d)            # deviates: own draw, pool off for the rest of the step
    rest = Fn.droppath_scale(1, 16, 0.5, d)
    Fn.droppath_end_step()
    assert odd.shape == (3, 16) and rest.untyped_storage().data_ptr() != first.untyped_storage().data_ptr()
    Fn.droppath_begin_step(d)                          # the deviating sequence is the new plan
    again = [Fn.droppath_scale(1, 16, 0.9, d), Fn.droppath_scale(3, 16, 0.8, d), Fn.droppath_scale(1, 16, 0.5, d)]
    Fn.droppath_end_step()
    assert len({t.untyped_storage().data_ptr() for t in again}) == 1
    Fn.droppath_reset()
