#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REAL REFERENCE.

Run only in the build container (``/root/reference`` mounted):

    python tests/golden/make_golden.py

Nothing from the reference is copied: the script imports
``/root/reference/models/{lora,swin_transformer_mtlora,swin_mtl}.py`` and
``/root/reference/mtl_loss_schemes.py`` (with the three stub modules of SURVEY.md
Appendix A standing in for timm / termcolor / ptflops, which are not installed),
runs them on seeded inputs and stores *inputs, parameters, outputs and gradients*
as small ``.pt`` files.  Large models are filled with ``oracle.det_fill_`` (a
name-keyed deterministic generator) so only checksums / samples are stored.
"""
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mtlora_oracle as O  # noqa: E402  (det_fill_ / config helpers only)

REF = "/root/reference"


def _install_stubs():
    class DropPath(nn.Module):  # timm==0.9.2 DropPath restated
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0 and self.scale_by_keep:
                m.div_(keep)
            return x * m

    tl = types.ModuleType("timm.models.layers")
    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules.update({"timm": types.ModuleType("timm"), "timm.models": types.ModuleType("timm.models"),
                        "timm.models.layers": tl})
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules["termcolor"] = tc
    pf = types.ModuleType("ptflops")
    pf.get_model_complexity_info = lambda *a, **k: (None, None)
    sys.modules["ptflops"] = pf


def _import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    import importlib
    lora = importlib.import_module("models.lora")
    swin = importlib.import_module("models.swin_transformer_mtlora")
    mtl = importlib.import_module("models.swin_mtl")
    losses = importlib.import_module("mtl_loss_schemes")
    return lora, swin, mtl, losses


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} kB")


def checksum(t: torch.Tensor):
    """size-independent summary: sum, abs-sum and 16 strided samples (fp64)."""
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return {"shape": tuple(t.shape), "sum": f.sum().item(), "abssum": f.abs().sum().item(),
            "idx": idx, "samples": f[idx].clone()}


# ----------------------------------------------------------------------------
def gen_linear(lora):
    """MTLoRALinear fwd + grads: every shared_mode x {tasks None, tasks, x_tasks}, bias on/off,
    r_t != r_s (lora.py:159-284)."""
    tasks = ["semseg", "normals", "sal", "human_parts"]
    K, N, M = 40, 56, 18
    cases = {}
    spec = [
        ("matrix_notasks", "matrix", None, False, True, {"shared": 16}),
        ("matrix_tasks", "matrix", tasks, False, True, {"shared": 16, **{t: 4 for t in tasks}}),
        ("matrix_xtasks", "matrix", tasks, True, True, {"shared": 16, **{t: 4 for t in tasks}}),
        ("matrix_xtasks_nobias_r", "matrix", tasks, True, False,
         {"shared": 8, "semseg": 4, "normals": 8, "sal": 12, "human_parts": 16}),
        ("matrixv2_xtasks", "matrixv2", tasks, True, True, {"shared": 16, **{t: 4 for t in tasks}}),
        ("matrixv2_tasks", "matrixv2", tasks, False, True, {"shared": 16, **{t: 4 for t in tasks}}),
        ("addition_xtasks", "addition", tasks, True, True, {"shared": 16, **{t: 4 for t in tasks}}),
        ("r0", "matrix", None, False, True, {"shared": 0}),
    ]
    for name, mode, tk, use_xt, bias, r in spec:
        torch.manual_seed(0)
        scale_t = {t: 4.0 - 0.5 * i for i, t in enumerate(tasks)} if tk else 1.0
        m = lora.MTLoRALinear(K, N, r=r, lora_shared_scale=2.5, lora_task_scale=scale_t, lora_dropout=0.0,
                              tasks=tk, shared_mode=mode, bias=bias).double()
        O.det_fill_(m.named_parameters())
        m = m.double()
        x = O.det_tensor(f"{name}.x", (2, M // 2, K), 1.0).double().requires_grad_(True)
        xt = None
        if use_xt:
            xt = {t: O.det_tensor(f"{name}.x.{t}", (2, M // 2, K), 1.0).double().requires_grad_(True) for t in tk}
        y, yt = m(x, xt)
        gy = O.det_tensor(f"{name}.gy", y.shape, 1.0).double()
        loss = (y * gy).sum()
        gyt = {}
        if yt is not None:
            for t in tk:
                gyt[t] = O.det_tensor(f"{name}.gy.{t}", y.shape, 1.0).double()
                loss = loss + (yt[t] * gyt[t]).sum()
        loss.backward()
        cases[name] = {
            "mode": mode, "tasks": tk, "r": r, "bias": bias, "scale_s": 2.5, "scale_t": scale_t,
            "params": {n: p.detach().clone() for n, p in m.named_parameters()},
            "x": x.detach().clone(), "x_tasks": {t: v.detach().clone() for t, v in xt.items()} if xt else None,
            "y": y.detach().clone(), "y_tasks": {t: v.detach().clone() for t, v in yt.items()} if yt else None,
            "gy": gy, "gy_tasks": gyt,
            "dx": x.grad.clone(), "dx_tasks": {t: v.grad.clone() for t, v in xt.items()} if xt else None,
            "grads": {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
        }
    save("linear.pt", cases)


def gen_window(swin):
    """roll + window_partition / window_reverse + roll (swin_transformer_mtlora.py:84-116, 336-386;
    same oracle as kernels/window_process/unit_test.py:96-115)."""
    out = {}
    for name, (B, H, W, C, ws, shift) in {"sq": (2, 14, 14, 8, 7, 3), "rect": (1, 14, 21, 6, 7, 2),
                                          "noshift": (2, 7, 14, 4, 7, 0), "unit": (2, 28, 28, 8, 7, 2)}.items():
        x = O.det_tensor(f"win.{name}", (B, H, W, C), 1.0)
        xs = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2)) if shift > 0 else x
        part = swin.window_partition(xs, ws)
        w = O.det_tensor(f"win.{name}.w", part.shape, 1.0)
        rev = swin.window_reverse(w, ws, H, W)
        merged = torch.roll(rev, shifts=(shift, shift), dims=(1, 2)) if shift > 0 else rev
        out[name] = {"dims": (B, H, W, C, ws, shift), "x": x, "partitioned": part, "w": w, "merged": merged}
    save("window_ops.pt", out)


def gen_attention(swin):
    """WindowAttention fwd + grads, mask / no mask (swin_transformer_mtlora.py:119-227)."""
    tasks = ["semseg", "normals"]
    mt = O.mtlora_config(tasks, r_shared=8, r_task=4, dropout=0.0)
    out = {}
    for name, use_mask, lora in (("nomask", False, False), ("mask", True, True)):
        torch.manual_seed(0)
        dim, heads, ws = 64, 2, 7
        att = swin.WindowAttention(dim, (ws, ws), heads, lora=lora, tasks=tasks, mtlora=mt, layer_idx=0).double()
        O.det_fill_(att.named_parameters())
        att = att.double().eval()
        mask = O.shifted_window_mask(14, 14, ws, 3).double() if use_mask else None
        x = O.det_tensor(f"att.{name}.x", (4, 49, dim), 1.0).double().requires_grad_(True)
        y, yt = att(x, mask)
        gy = O.det_tensor(f"att.{name}.gy", y.shape, 1.0).double()
        loss = (y * gy).sum()
        if yt is not None:
            for t in tasks:
                loss = loss + (yt[t] * O.det_tensor(f"att.{name}.gy.{t}", y.shape, 1.0).double()).sum()
        loss.backward()
        # the attention core alone (qkv -> attn output before proj), via hooks-free recomputation
        with torch.no_grad():
            qkv, _ = att.qkv(x)
        out[name] = {
            "lora": lora, "tasks": tasks, "heads": heads, "ws": ws, "mask": mask,
            "params": {n: p.detach().clone() for n, p in att.named_parameters()},
            "rel_index": att.relative_position_index.clone(),
            "x": x.detach().clone(), "qkv": qkv.clone(), "y": y.detach().clone(),
            "y_tasks": {t: v.detach().clone() for t, v in yt.items()} if yt else None,
            "dx": x.grad.clone(),
            "grads": {n: p.grad.clone() for n, p in att.named_parameters() if p.grad is not None},
        }
    save("window_attention.pt", out)


def gen_block(swin):
    """SwinTransformerBlock (lora=True, shift>0 and shift=0), eval mode (:326-408)."""
    tasks = ["semseg", "sal"]
    mt = O.mtlora_config(tasks, r_shared=8, r_task=4, dropout=0.0)
    out = {}
    for name, shift, lora in (("shift_lora", 3, True), ("noshift_plain", 0, False)):
        blk = swin.SwinTransformerBlock(64, (14, 14), 2, window_size=7, shift_size=shift, lora=lora, tasks=tasks,
                                        mtlora=mt, layer_idx=0, drop_path=0.1)
        O.det_fill_(blk.named_parameters())
        blk = blk.double().eval()
        x = O.det_tensor(f"blk.{name}.x", (1, 196, 64), 1.0).double().requires_grad_(True)
        y, yt = blk(x)
        loss = (y * O.det_tensor(f"blk.{name}.gy", y.shape, 1.0).double()).sum()
        if yt is not None:
            for t in tasks:
                loss = loss + (yt[t] * O.det_tensor(f"blk.{name}.gy.{t}", y.shape, 1.0).double()).sum()
        loss.backward()
        out[name] = {
            "shift": shift, "lora": lora, "tasks": tasks,
            "x": x.detach().clone(), "y": y.detach().clone(),
            "y_tasks": {t: v.detach().clone() for t, v in yt.items()} if yt else None,
            "dx": x.grad.clone(),
            "grads": {n: (p.grad.clone() if p.numel() <= 2048 else checksum(p.grad))
                      for n, p in blk.named_parameters() if p.grad is not None},
            "param_names": [n for n, _ in blk.named_parameters()],
        }
    save("swin_block.pt", out)


def _ref_model(swin, mtl, cfg, num_outputs):
    mt = cfg["mtlora"]
    bb = swin.SwinTransformerMTLoRA(img_size=cfg["img_size"], patch_size=4, in_chans=3, num_classes=0,
                                    embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                    window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                                    drop_path_rate=cfg["drop_path_rate"], ape=False, norm_layer=nn.LayerNorm,
                                    patch_norm=True, use_checkpoint=False, fused_window_process=False,
                                    tasks=cfg["tasks"], mtlora=mt)
    mc = O.Cfg(TASKS=cfg["tasks"],
               TASKS_CONFIG=O.Cfg(ALL_TASKS=O.Cfg(NUM_OUTPUT=num_outputs)),
               MODEL=O.Cfg(MTLORA=mt, DECODER_HEAD=O.Cfg({t: "hrnet" for t in cfg["tasks"]}),
                           DECODER_CHANNELS=[18, 36, 72, 144], DECODER_DOWNSAMPLER=True,
                           PER_TASK_DOWNSAMPLER=True))
    model = mtl.MultiTaskSwin(bb, mc)
    return model


def gen_backbone_small(swin):
    """2-stage backbone (56x56 image -> 14x14, 7x7 tokens), T=2, r_s=8, r_t=4, eval; full outputs stored."""
    tasks = ["semseg", "normals"]
    cfg = O.swin_t_cfg(img_size=56, tasks=tasks, r_shared=8, r_task=4, depths=(2, 2), num_heads=(3, 6),
                       drop_path_rate=0.1, dropout=0.05)
    bb = swin.SwinTransformerMTLoRA(img_size=56, patch_size=4, in_chans=3, num_classes=0, embed_dim=96,
                                    depths=[2, 2], num_heads=[3, 6], window_size=7, drop_path_rate=0.1,
                                    tasks=tasks, mtlora=cfg["mtlora"])
    O.det_fill_(bb.named_parameters())
    bb = bb.double().eval()
    x = O.det_tensor("bbs.x", (1, 3, 56, 56), 1.0).double()
    stages = bb(x, return_stages=True)
    loss = 0
    for i, (s, tl) in enumerate(stages):
        loss = loss + (s * O.det_tensor(f"bbs.g.{i}", s.shape, 1.0).double()).sum()
        for t in tasks:
            loss = loss + (tl[t] * O.det_tensor(f"bbs.g.{i}.{t}", s.shape, 1.0).double()).sum()
    loss.backward()
    keep = ["layers.0.blocks.1.attn.proj.lora_tasks_A.semseg", "layers.0.blocks.0.attn.qkv.lora_shared_B",
            "layers.1.blocks.1.mlp.fc1.lora_tasks_B.normals", "layers.0.blocks.1.attn.relative_position_bias_table",
            "layers.0.blocks.0.norm1.weight", "layers.0.downsample.reduction.weight", "patch_embed.proj.weight",
            "layers.1.blocks.1.mlp.fc2.lora_shared_A", "layers.1.blocks.0.mlp.fc2.lora_shared_A"]
    grads = dict(bb.named_parameters())
    save("backbone_small.pt", {
        "tasks": tasks, "names": list(bb.state_dict().keys()),
        "stages": [(s.detach().clone(), {t: v.detach().clone() for t, v in tl.items()}) for s, tl in stages],
        "grads": {k: (None if grads[k].grad is None else
                      (grads[k].grad.clone() if grads[k].numel() <= 4096 else checksum(grads[k].grad))) for k in keep},
        "grad_is_none": sorted(n for n, p in bb.named_parameters() if p.grad is None),
    })


def gen_backbone_options(swin, lora):
    """the MTLoRA options the shipped yamls switch on beyond the defaults, each on a 2-stage backbone (56 px, T=2, r 8/4,
    eval mode, fp64): DOWNSAMPLER_ENABLED (the six mtlora_plus_* yamls: PatchMerging.reduction becomes an MTLoRALinear,
    swin_transformer_mtlora.py:442-447), INTERMEDIATE_SPECIALIZATION (every block specialises, :60-64 / :131-137) and
    TRAINABLE_SCALE_SHARED (lora.py:205-209: the shared scale is a 1-element Parameter with its own gradient).
    (TRAINABLE_SCALE_PER_TASK crashes in the reference itself with the dict the backbone passes, lora.py:212 -- SURVEY a2.)"""
    tasks = ["semseg", "normals"]
    out = {}
    for name, over in (("downsampler", dict(DOWNSAMPLER_ENABLED=True)),
                       ("intermediate", dict(INTERMEDIATE_SPECIALIZATION=True)),
                       ("trainable_scale", dict(TRAINABLE_SCALE_SHARED=True))):
        cfg = O.swin_t_cfg(img_size=56, tasks=tasks, r_shared=8, r_task=4, depths=(2, 2), num_heads=(3, 6),
                           drop_path_rate=0.1, dropout=0.05, **over)
        bb = swin.SwinTransformerMTLoRA(img_size=56, patch_size=4, in_chans=3, num_classes=0, embed_dim=96,
                                        depths=[2, 2], num_heads=[3, 6], window_size=7, drop_path_rate=0.1,
                                        tasks=tasks, mtlora=cfg["mtlora"])
        O.det_fill_(bb.named_parameters())
        lora.mark_only_lora_as_trainable(bb, bias="none")
        bb = bb.double().eval()
        x = O.det_tensor("bbo.x", (1, 3, 56, 56), 1.0).double()
        stages = bb(x, return_stages=True)
        loss = 0
        for i, (s, tl) in enumerate(stages):
            loss = loss + (s * O.det_tensor(f"bbo.g.{i}", s.shape, 1.0).double()).sum()
            for t in tasks:
                loss = loss + (tl[t] * O.det_tensor(f"bbo.g.{i}.{t}", s.shape, 1.0).double()).sum()
        loss.backward()
        named = dict(bb.named_parameters())
        out[name] = {
            "over": over, "tasks": tasks, "names": list(bb.state_dict().keys()),
            "trainable": [n for n, p in named.items() if p.requires_grad],
            "stages": [(s.detach().clone(), {t: v.detach().clone() for t, v in tl.items()}) for s, tl in stages],
            "grads": {n: (p.grad.clone() if p.numel() <= 1024 else checksum(p.grad)) for n, p in named.items()
                      if p.requires_grad and p.grad is not None},
            "grad_is_none": sorted(n for n, p in named.items() if p.requires_grad and p.grad is None),
        }
    save("backbone_options.pt", out)


def gen_c1(lora, swin, mtl, losses):
    """BASELINE config C1: Swin-T/224, 1 task (semseg), r=4, bs=2, CPU fwd+bwd; eval-mode dropout
    (main.py:329-354 step without the optimizer).  Stores loss, checksums and grad slices."""
    tasks = ["semseg"]
    cfg = O.swin_t_cfg(img_size=224, tasks=tasks, r_shared=4, r_task=4, drop_path_rate=0.2)
    model = _ref_model(swin, mtl, cfg, {"semseg": 21})
    O.det_fill_(list(model.named_parameters()) + list(model.named_buffers()))
    lora.mark_only_lora_as_trainable(model.backbone, bias="none", freeze_patch_embed=False, freeze_norm=False,
                                     free_relative_bias=False, freeze_downsample_reduction=False)
    model.eval()
    img, tg = O.synthetic_batch(2, 224, tasks, seed=1234)
    crit = losses.MultiTaskLoss(tasks, nn.ModuleDict({t: losses.get_loss({}, t) for t in tasks}),
                                {"semseg": 1.0})
    out = model(img)
    loss, _ = crit(out, tg)
    loss.backward()
    named = dict(model.named_parameters())
    keep = ["backbone.layers.0.blocks.1.attn.proj.lora_tasks_B.semseg",
            "backbone.layers.2.blocks.5.mlp.fc1.lora_tasks_A.semseg",
            "backbone.layers.1.blocks.0.attn.qkv.lora_shared_A",
            "backbone.layers.3.blocks.1.attn.relative_position_bias_table",
            "backbone.layers.2.blocks.3.norm2.weight", "backbone.patch_embed.norm.bias",
            "downsampler.semseg.downsample_2.weight", "decoders.decoders.semseg.last_layer.3.bias"]
    n_params = sum(p.numel() for p in model.parameters())
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_lora = sum(p.numel() for n, p in model.named_parameters() if "lora_" in n)
    print(f"  C1: params {n_params}, trainable {n_train}, lora {n_lora}, loss {loss.item():.6f}")
    save("c1_model.pt", {
        "loss": loss.item(), "out": checksum(out["semseg"]), "n_params": n_params, "n_trainable": n_train,
        "n_lora": n_lora, "state_names": list(model.state_dict().keys()),
        "trainable_names": [n for n, p in model.named_parameters() if p.requires_grad],
        "grads": {k: checksum(named[k].grad) for k in keep},
        "grad_is_none": sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None),
    })


def gen_c2_names(lora, swin, mtl):
    """C2-shaped model structure facts: names, counts, trainable set."""
    tasks = ["semseg", "normals", "sal", "human_parts"]
    cfg = O.swin_t_cfg(img_size=448, tasks=tasks, r_shared=64, r_task=4)
    model = _ref_model(swin, mtl, cfg, {t: O.NUM_OUTPUT[t] for t in tasks})
    lora.mark_only_lora_as_trainable(model.backbone, bias="none")
    save("c2_structure.pt", {
        "state": {k: tuple(v.shape) for k, v in model.state_dict().items()},
        "trainable": [n for n, p in model.named_parameters() if p.requires_grad],
        "n_params": sum(p.numel() for p in model.parameters()),
        "n_trainable": sum(p.numel() for p in model.parameters() if p.requires_grad),
    })


def gen_losses(losses):
    """mtl_loss_schemes.py losses on small synthetic maps."""
    out = {}
    img, tg = O.synthetic_batch(2, 16, ["semseg", "normals", "sal", "human_parts", "depth"], seed=7)
    for t in tg:
        pred = O.det_tensor(f"loss.{t}", (2, O.NUM_OUTPUT[t], 16, 16), 1.0).requires_grad_(True)
        crit = losses.get_loss({}, t)
        l = crit(pred, tg[t])
        l.backward()
        out[t] = {"pred": pred.detach().clone(), "label": tg[t], "loss": l.item(), "dpred": pred.grad.clone()}
    save("losses.pt", out)


def gen_checkpoint(lora, swin):
    """reference utils.load_checkpoint (utils.py:41-176) on a synthetic VANILLA Swin checkpoint -> MTLoRA backbone: key
    mapping (.weight -> .linear.weight), attn_mask strip, relative-position table re-interpolation (window 7 -> 4).
    Stores only names / shapes of the synthetic checkpoint (values = det_tensor(name)) and checksums of the loaded model."""
    import importlib
    import tempfile
    for name in ("cv2", "imageio"):  # imported by utils.py for its image helpers; absent here, unused on this path
        sys.modules.setdefault(name, types.ModuleType(name))
    utils = importlib.import_module("utils")
    vanilla_mod = importlib.import_module("models.swin_transformer")
    van = vanilla_mod.SwinTransformer(img_size=112, patch_size=4, in_chans=3, num_classes=0, embed_dim=48, depths=[2, 2],
                                      num_heads=[2, 4], window_size=7, drop_path_rate=0.0)
    sd = {k: v.clone() for k, v in van.state_dict().items()}
    for k, v in sd.items():
        if torch.is_floating_point(v) and "attn_mask" not in k:
            v.copy_(O.det_tensor("ckpt." + k, v.shape, 0.05))
    tasks = ["semseg", "normals"]
    mt = O.mtlora_config(tasks, r_shared=8, r_task=4, n_stages=2, SPLIT_QKV=False)
    tgt = swin.SwinTransformerMTLoRA(img_size=64, patch_size=4, in_chans=3, num_classes=0, embed_dim=48, depths=[2, 2],
                                     num_heads=[2, 4], window_size=4, drop_path_rate=0.0, tasks=tasks, mtlora=mt)
    O.det_fill_(tgt.named_parameters())
    before = {k: v.clone() for k, v in tgt.state_dict().items()}

    class Log:
        def __init__(self):
            self.w = []

        def info(self, m):
            pass

        def warning(self, m):
            self.w.append(str(m))

    log = Log()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "vanilla.pth")
        torch.save({"model": sd}, path)
        cfg = O.Cfg(MODEL=O.Cfg(RESUME="", RESUME_BACKBONE=path, MTLORA=mt, UPDATE_RELATIVE_POSITION=True),
                    TRAIN=O.Cfg(SKIP_DECODER_CKPT=False), EVAL_MODE=True)
        utils.load_checkpoint(cfg, tgt, None, None, None, log, backbone=True)
    after = tgt.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    save("checkpoint_map.pt", {
        "ckpt": {k: (tuple(v.shape), str(v.dtype)) for k, v in sd.items()},
        "warnings": log.w,
        "changed": changed,
        "loaded": {k: checksum(after[k]) for k in changed},
        "tables": {k: after[k].clone() for k in changed if "relative_position_bias_table" in k},
    })


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted; golden vectors can only be regenerated in the build container"
    torch.set_num_threads(8)
    lora, swin, mtl, losses = _import_reference()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints its config in ctors
        pass
    gen_linear(lora)
    gen_window(swin)
    gen_attention(swin)
    gen_block(swin)
    gen_losses(losses)
    gen_backbone_small(swin)
    gen_c2_names(lora, swin, mtl)
    gen_c1(lora, swin, mtl, losses)
    gen_checkpoint(lora, swin)
    gen_backbone_options(swin, lora)
