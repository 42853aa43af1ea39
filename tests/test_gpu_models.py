"""Model-level parity of the HIP path for every BASELINE.json configuration (run on an MI355X: -m gpu).

  * C1 (configs[0]: Swin-T/224, 1 task, r=4, bs=2) against the fixture captured from the REAL reference
    (tests/golden/c1_model.pt: loss 3.370951, output checksum, 8 gradient checksums, the grad-is-None set);
  * C2 (Swin-T, 4 tasks, r 64/4), C4 (Swin-B, r=128) and C5 (8 synthetic tasks, r=4 and r=256) as whole
    models -- full depth, train-mode statistics -- against the oracle's full model evaluated in fp64, in fp32
    (north star tolerance 1e-3) and under bf16 autocast with the per-task head streams on (1e-2 on loss and
    outputs; gradients through 12-24 bf16 blocks are held to max(1e-2, 2 x the error the reference's OWN eager
    bf16-autocast dataflow makes against the same fp64 values) -- measured in the same test, on the same inputs).

The oracle runs through ATen in fp64 on the same GPU (it is device-agnostic plain PyTorch): a Swin-B fp64
forward+backward takes minutes on host cores and seconds there.  Dropout / DropPath are off in the model tests
(both sides deterministic); their generators are tested element-for-element in test_gpu_kernels.py.
"""
import pytest
import torch

from oracle import mtlora_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _hip_loss(model, crit, img, tg, amp, concurrent=None):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        return crit.combine(model(img, upsample=False, per_task_fn=lambda t, lo: crit.task_low(t, lo, tg[t]),
                                  concurrent=concurrent))


def _oracle_run(sd, trainable, img, tg, cfg, tasks, dtype, amp, train, device=None, rng=None):
    """oracle.full_model + multi_task_loss in `dtype` (fp64 = the reference values; fp32 under bf16 autocast = the
    reference's own eager reduced-precision path); returns loss, per-task losses, {name: grad}.  Runs through ATen on the
    GPU; if this ROCm build lacks an fp64 kernel for one of the ops, the same code runs on the host cores instead."""
    device = device or dev()
    try:
        P = {k: v.detach().to(device=device, dtype=dtype).clone() for k, v in sd.items() if v.is_floating_point()}
        for k in P:
            if k in trainable:
                P[k].requires_grad_(True)
        x = img.to(device=device, dtype=dtype)
        tgt = {t: v.to(device=device, dtype=dtype) for t, v in tg.items()}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = O.full_model(P, x, cfg, train=train, rng=rng if rng is not None else torch.Generator().manual_seed(0))
            loss, per = O.multi_task_loss({k: v.float() if amp else v for k, v in out.items()}, tgt, tasks)
        loss.backward()
    except RuntimeError:
        if device.type == "cpu" or amp:
            raise
        return _oracle_run(sd, trainable, img, tg, cfg, tasks, dtype, amp, train, torch.device("cpu"), rng)
    return (loss.detach().cpu(), {t: v.detach().cpu() for t, v in per.items()},
            {k: (None if P[k].grad is None else P[k].grad.cpu()) for k in trainable})


def _grad_errors(grads, ref, skip=()):
    """max |g - r| / max(|r|max, floor) per parameter ({name: grad} both sides); floor = 1e-6 of the largest gradient
    magnitude in the model."""
    gmax = max(r.abs().max().item() for r in ref.values() if r is not None)
    errs = {}
    for n, r in ref.items():
        g = grads[n]
        if r is None:
            assert g is None, f"{n}: gradient where the reference has none"
            continue
        assert g is not None, f"{n}: no gradient"
        if n in skip:
            continue
        scale = max(r.abs().max().item(), 1e-6 * gmax)
        errs[n] = (g.double().cpu() - r.double().cpu()).abs().max().item() / scale
    return errs, gmax


# parameter families whose fp32 gradient may move by more than the per-tensor tolerance because ONE forward value rounds across a
# kink (see test_config_model_vs_oracle): the heads' own convolutions / BatchNorm (the ReLU sits right behind them) and the
# per-task downsamplers that feed them, and the TASK-SPECIFIC low-rank factors, whose whole gradient arrives through that one head
# (c5:4 fp32: `lora_tasks_*.t2` at 3.5e-3 .. 5e-3 behind `decoders.t2.last_layer.0.weight` at 2e-2, eager fp32 at 1e-5: the same
# flipped element).  Everything shared between the tasks (shared factors, norms, tables, patch embedding) has to meet the tolerance.
KINK_AFFECTED = ("decoders.decoders.", "downsampler.", "lora_tasks_")

MODEL_CASES = {
    # name: (config row, overrides)
    "c2_swin_t_r64_4": ("c2", {}),
    "c4_swin_b_r128": ("c4", {}),
    "c5_8task_r4": ("c5:4", {}),
    "c5_8task_r16": ("c5:16", {}),
    "c5_8task_r64": ("c5:64", {}),
    "c5_8task_r256": ("c5:256", {}),
    # the MFMA-balance config at the BASELINE resolution (448 px: stage-0 M = 25 088 rows at B = 2, the shapes bench.py's c4 leg
    # launches per image) -- VERDICT r05 item 8; [auto] only (the 224 px case above runs in every family)
    "c4_swin_b_r128_448px": ("c4", {"_img": 448}),
}


@pytest.mark.parametrize("amp", [False, True], ids=["fp32", "bf16_streams"])
@pytest.mark.parametrize("case", list(MODEL_CASES))
def test_config_model_vs_oracle(case, amp):
    """whole model of a BASELINE config (full depth, 224 px, B=2, train mode => batch-statistics BatchNorm, the deferred
    residual + PatchMerging kernels, fused GELU, fused losses) vs the oracle in fp64."""
    from mtlora_amd import mtl_harness as H
    name, over = MODEL_CASES[case]
    over = dict(over)
    px = over.pop("_img", 224)
    row = H.config(name)
    tasks = list(row["tasks"])
    model = H.build_config_model(name, seed=3, img_size=px, drop_path_rate=0.0, DROPOUT=[0.0] * 4, **over).to(dev())
    _condition_normals_heads(model, tasks)
    model.train()
    crit = H.MultiTaskLoss(tasks)
    img, tg = H.synthetic_batch(2, px, tasks, seed=5, device=dev())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}  # before the forward updates the BN running stats
    loss, per = _hip_loss(model, crit, img, tg, amp, concurrent=True if amp else False)
    loss.backward()
    torch.cuda.synchronize()

    cfg = O.swin_t_cfg(img_size=px, tasks=tasks, r_shared=row["r_shared"], r_task=row["r_task"], embed_dim=row["embed_dim"],
                       depths=row["depths"], num_heads=row["num_heads"], drop_path_rate=0.0, dropout=0.0)
    _compare_with_oracle(case, amp, model, sd, loss, per, img, tg, cfg, tasks)


def _compare_with_oracle(case, amp, model, sd, loss, per, img, tg, cfg, tasks, rng=None):
    """loss, per-task losses and every trainable gradient of ``model`` (already run: ``loss.backward()`` done) against the oracle in
    fp64 on the state dict ``sd`` the run started from, calibrated by the oracle's own run at the model's precision (see
    test_config_model_vs_oracle).  ``rng``: replayed randomness for train-mode dropout / DropPath (oracle ``_lin`` / ``_drop_path``)."""
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    rl, rper, rg = _oracle_run(sd, trainable, img, tg, cfg, tasks, torch.float64, False, True, rng=rng)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    bn_bias = {n for n in trainable if n.endswith("last_layer.0.bias")}  # in front of a BatchNorm: gradient analytically 0
    errs, gmax = _grad_errors(grads, rg, skip=bn_bias)
    for n in bn_bias:
        assert grads[n].abs().max().item() <= 1e-3 * gmax and rg[n].abs().max().item() <= 1e-3 * gmax, n
    tol = 1e-2 if amp else 1e-3
    assert abs(loss.item() - rl.item()) <= tol * abs(rl.item()), (loss.item(), rl.item())
    for t in tasks:
        assert abs(per[t].item() - rper[t].item()) <= tol * max(1.0, abs(rper[t].item())), (t, per[t].item(), rper[t].item())
    assert len(errs) > 200
    # calibration: the reference's OWN eager dataflow at the same precision (fp32, or fp32 parameters under bf16 autocast) on
    # the same parameters / batch, against the same fp64 values.  A gradient the eager path itself only resolves to x % is
    # held to max(floor, 2x) -- never looser than what the reference delivers, never tighter than its own rounding noise.
    el, _, eg = _oracle_run(sd, trainable, img, tg, cfg, tasks, torch.float32, amp, True, rng=rng)
    eerrs, _ = _grad_errors(eg, rg, skip=bn_bias)
    assert abs(el.item() - rl.item()) <= (5e-2 if amp else 1e-3) * abs(rl.item())  # sanity of the calibration run itself
    floor = 1e-2 if amp else 5e-3
    bad = {n: (e, eerrs[n]) for n, e in errs.items() if e > max(floor, 2.0 * eerrs[n])}
    ratios = sorted(errs[n] / max(eerrs[n], 1e-12) for n in errs if eerrs[n] > floor / 10)
    med, emed = sorted(errs.values())[len(errs) // 2], sorted(eerrs.values())[len(eerrs) // 2]
    _report(case, amp, dict(n=len(errs), med=med, eager_med=emed, max=max(errs.values()), eager_max=max(eerrs.values()),
                            ratio_med=ratios[len(ratios) // 2] if ratios else None,
                            ratio_p90=ratios[int(0.9 * len(ratios))] if ratios else None, n_bad=len(bad),
                            worst=sorted(bad.items(), key=lambda kv: -kv[1][0])[:3]))
    # Acceptance.  Gradients of this model are not smooth functions of the arithmetic: the heads' BatchNorm + ReLU and the
    # L1 / normalisation of NormalsLoss have kinks, and ONE element of 1.7 M whose pre-activation rounds to the other side of
    # 0 moves a BatchNorm beta gradient by 1e-3 and one entry of dx by 2e-2 (measured: tools/debug_bn.py; the eager path
    # flips other elements).  So single tensors are held to a cap that still catches a wrong formula (those give O(1)), and
    # the population is held tight:
    if not amp:
        # typical tensor at the north-star fp32 tolerance -- or at what the reference's own fp32 eager path resolves when the kinks
        # dominate the whole population (c5:256: eager median 1.2e-3, HIP 1.6e-3; kink-free probe below: 5e-6)
        assert med <= max(1e-3, 1.5 * emed), (med, emed)
        # per tensor: the north-star 1e-3 with k = 5 of head-room (c5:256, eight heads' worth of kinks and rank-256 sums: one shared
        # LayerNorm weight at 4.5e-3 with the population median at 1.6e-3 and the eager path's at 1.2e-3), or twice what the
        # reference's own fp32 eager path resolves.  Only the KINK_AFFECTED families may exceed it (and stay capped): their
        # gradients are piecewise constant in a pre-activation (ReLU behind the heads' BatchNorm, |.| and 1 / ||.|| in NormalsLoss), so
        # one element rounding across the kink moves them by percents in fp32 on EITHER path -- test_config_model_kinkfree_probe
        # pins that those same tensors agree to 1e-3 once the kinks are taken out.
        over = {n: e for n, e in errs.items() if e > max(5e-3, 2.0 * eerrs[n])}
        stray = {n: e for n, e in over.items() if not any(k in n for k in KINK_AFFECTED)}
        assert not stray, sorted(stray.items(), key=lambda kv: -kv[1])[:5]
        assert max(errs.values()) <= max(5e-2, 2.0 * max(eerrs.values())), sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        assert len(over) <= 0.10 * len(errs), len(over)
        return
    # bf16: in aggregate the HIP path must be as accurate as the reference's own eager bf16-autocast path ...
    assert med <= max(floor, 1.25 * emed), (med, emed)
    assert ratios and ratios[len(ratios) // 2] <= 1.25, ratios[len(ratios) // 2]
    # ... 97 % of the tensors within 2x of it, and none beyond 8x (tensors the eager path itself resolves to 5-50 % are two
    # draws of the same rounding noise)
    assert len(bad) <= 0.03 * len(errs), sorted(bad.items(), key=lambda kv: -kv[1][0])[:5]
    # (the cap skips tensors of < 64 elements -- the 3-element biases of the regression heads are sums over every pixel with
    # near-total cancellation: a single noisy number per side; the eager value itself moves 2x between runs, its reductions
    # use atomics)
    worst = {n: v for n, v in bad.items() if v[0] > max(2e-2, 8.0 * v[1]) and grads[n].numel() >= 64}
    assert not worst, sorted(worst.items(), key=lambda kv: -kv[1][0])[:5]


class _ReplayRandomness:
    """the dropout masks and DropPath factors the HIP model drew in ONE forward, handed to the oracle (``rng`` of oracle._lin /
    oracle._drop_path): masks from the specified generator (oracle.dropout_keep_mask_t of each layer's recorded seed, keyed by the
    row index in IMAGE order -- the HIP path never leaves that order -- and permuted into the reference's window order for qkv /
    proj), DropPath factors as recorded from functional.droppath_scale."""

    def __init__(self, seeds, factors, blocks, tasks, p):
        self.seeds, self.factors, self.blocks, self.tasks, self.p = seeds, factors, blocks, list(tasks), p

    def keep_mask(self, pre, x):
        blk = pre.rsplit(".", 2)[0]
        B, H, W, ws, shift = self.blocks[blk]
        K = x.shape[-1]
        m = O.dropout_keep_mask_t(self.seeds[pre], 0, B * H * W, K, self.p, device=x.device)
        if ".attn." in pre:  # the reference runs qkv / proj on window-ordered tokens (swin_transformer_mtlora.py:336-353)
            m = O.roll_and_window_partition(m.view(B, H, W, K), shift, ws).reshape(-1, ws * ws, K)
        return m.reshape(x.shape)

    def droppath(self, tag, x):
        blk, which, t = tag
        f = self.factors[(blk, which)]
        return f[0 if t is None else 1 + self.tasks.index(t)]


@pytest.mark.parametrize("amp", [False, True], ids=["fp32", "bf16_streams"])
def test_c2_train_mode_model_vs_oracle(amp):
    """BASELINE configs[1] as the benchmark runs it -- 448 px, TRAIN mode with LoRA dropout 0.05 and DropPath 0.2, the one-call
    blocks, the factor packer's signature bookkeeping, the DropPath pool, the per-task streams -- against the fp64 oracle fed with
    the SAME randomness: every layer's dropout seed and every residual's DropPath factors are recorded while the HIP model runs,
    the oracle rebuilds the masks with the specified generator (the composition of the per-layer seeds across
    48 layers was only ever compared HIP-vs-HIP)."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    row = H.config("c2")
    tasks = list(row["tasks"])
    p_drop, dpr = 0.05, 0.2
    torch.manual_seed(17)
    model = H.build_config_model("c2", seed=3, img_size=448, drop_path_rate=dpr, DROPOUT=[p_drop] * 4).to(dev())
    _condition_normals_heads(model, tasks)
    model.train()
    crit = H.MultiTaskLoss(tasks)
    img, tg = H.synthetic_batch(2, 448, tasks, seed=5, device=dev())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    seeds, dps = [], []
    keep_ns, keep_dp = Fn.next_seed, Fn.droppath_scale

    def rec_seed():
        s_ = keep_ns()
        seeds.append(s_)
        return s_

    def rec_dp(n, B, keep, device):
        f = keep_dp(n, B, keep, device)
        dps.append(f.detach().clone())
        return f

    Fn.next_seed, Fn.droppath_scale = rec_seed, rec_dp
    try:
        loss, per = _hip_loss(model, crit, img, tg, amp, concurrent=True if amp else False)
    finally:
        Fn.next_seed, Fn.droppath_scale = keep_ns, keep_dp
    loss.backward()
    torch.cuda.synchronize()
    # which call drew what: four MTLoRALinear calls (qkv, proj, fc1, fc2) and -- where the block's drop_prob > 0 -- two residuals per
    # block, in model order
    bb = model.backbone
    seed_of, fac_of, blocks = {}, {}, {}
    si = di = 0
    for i, layer in enumerate(bb.layers):
        for j, blk in enumerate(layer.blocks):
            pre = f"backbone.layers.{i}.blocks.{j}"
            Hh, Ww = blk.input_resolution
            blocks[pre] = (2, Hh, Ww, blk.window_size, blk.shift_size)
            for nm in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                seed_of[f"{pre}.{nm}"] = seeds[si]
                si += 1
            if getattr(blk.drop_path, "drop_prob", 0.0) > 0.0:
                for which in ("attn", "mlp"):
                    f = dps[di]
                    fac_of[(pre, which)] = f if f.dim() == 2 else f.view(1, -1)
                    assert fac_of[(pre, which)].shape[0] == (1 + len(tasks) if blk.lora else 1), (pre, which, f.shape)
                    di += 1
    assert si == len(seeds) == 48 and di == len(dps) == 22, (si, len(seeds), di, len(dps))
    cfg = O.swin_t_cfg(img_size=448, tasks=tasks, r_shared=row["r_shared"], r_task=row["r_task"], embed_dim=row["embed_dim"],
                       depths=row["depths"], num_heads=row["num_heads"], drop_path_rate=dpr, dropout=p_drop)
    replay = _ReplayRandomness(seed_of, fac_of, blocks, tasks, p_drop)
    _compare_with_oracle("c2_train_448", amp, model, sd, loss, per, img, tg, cfg, tasks, rng=replay)


def _condition_normals_heads(model, tasks):
    """NormalsLoss divides the prediction by its L2 norm (mtl_loss_schemes.py:162-220): at random init the 3-channel outputs
    sit near 0, where d loss / d out ~ 1 / |out| turns rounding noise into O(1) gradient noise (the fp32 eager path then
    disagrees with fp64 by percents).  Parity is checked at a well-conditioned point instead: the final 1x1 conv's bias of
    every normals-kind head is moved away from 0 (|out| ~ 1), on both sides identically (it is part of the state dict)."""
    from mtlora_amd import mtl_harness as H
    with torch.no_grad():
        for t in tasks:
            if H.task_kind(t) == "normals":
                model.decoders.decoders[t].last_layer[3].bias.add_(torch.tensor([0.9, -0.7, 0.8], device=dev()))


def _report(case, amp, stats):
    """one line per model case into gpurun_out/ (pulled back from the GPU box): how close each precision runs to fp64"""
    import json
    import os
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/model_parity.jsonl", "a") as f:
            f.write(json.dumps({"case": case, "amp": amp, **{k: (v if not isinstance(v, list) else str(v)) for k, v in stats.items()}}) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("case", ["c5_8task_r4", "c5_8task_r256", "c2_swin_t_r64_4"])
def test_config_model_kinkfree_probe(case):
    """fp32 whole-model gradients with the KINKS taken out on both sides: heads' ReLU = identity, loss = a smooth quadratic probe of
    the upsampled outputs (no |.|, no 1 / ||.||, no argmax-like ignore masks).  Every gradient of the model is then a smooth
    function of the arithmetic and EVERY tensor must agree with the fp64 oracle to the north-star fp32 tolerance -- in
    particular the tensors that exceed it in test_config_model_vs_oracle (decoders.*.last_layer.0.weight at 1e-2 with the ReLU
    in place): a real bug in the rank-0 k_nt / k_tn plain_dW / BatchNorm-backward chain would show here as well."""
    from mtlora_amd import mtl_harness as H
    name, over = MODEL_CASES[case]
    row = H.config(name)
    tasks = list(row["tasks"])
    model = H.build_config_model(name, seed=3, img_size=224, drop_path_rate=0.0, DROPOUT=[0.0] * 4, **over).to(dev())
    for t in tasks:
        model.decoders.decoders[t].relu = False
    model.train()
    img, _ = H.synthetic_batch(2, 224, tasks, seed=5, device=dev())
    probes = {t: O.det_tensor("probe." + t, (1, H.num_output(t), 1, 1), 1.0).to(dev()) for t in tasks}
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def probe_loss(outs, cast):
        return sum((0.5 * (cast(outs[t]) * cast(probes[t])) ** 2).mean() + 0.1 * (cast(outs[t]) * cast(probes[t])).mean() for t in tasks)

    loss = probe_loss(model(img), lambda v: v.float())
    loss.backward()
    torch.cuda.synchronize()
    cfg = O.swin_t_cfg(img_size=224, tasks=tasks, r_shared=row["r_shared"], r_task=row["r_task"], embed_dim=row["embed_dim"],
                       depths=row["depths"], num_heads=row["num_heads"], drop_path_rate=0.0, dropout=0.0)
    cfg["head_relu"] = False
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    P = {k: v.detach().to(device=dev(), dtype=torch.float64).clone() for k, v in sd.items() if v.is_floating_point()}
    for k in P:
        if k in trainable:
            P[k].requires_grad_(True)
    rl = probe_loss(O.full_model(P, img.double(), cfg, train=True, rng=torch.Generator().manual_seed(0)), lambda v: v.double())
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 1e-3 * abs(rl.item()), (loss.item(), rl.item())
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    rg = {k: (None if P[k].grad is None else P[k].grad.cpu()) for k in trainable}
    bn_bias = {n for n in trainable if n.endswith("last_layer.0.bias")}
    errs, gmax = _grad_errors(grads, rg, skip=bn_bias)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    _report(case + ":kinkfree", False, dict(n=len(errs), med=sorted(errs.values())[len(errs) // 2], max=worst[0][1], worst=worst))
    assert worst[0][1] <= 1e-3, worst


@pytest.mark.parametrize("amp", [False, True], ids=["fp32", "bf16"])
def test_c1_reference_fixture_on_hip_path(golden, amp):
    """BASELINE configs[0] through the HIP kernels against the fixture produced by the real reference (eval mode, the
    reference's `main.py:329-354` forward + loss + backward without the optimizer): T = 1, r_s = r_t = 4 (two 4 -> 16
    padded rank segments), full 2-2-6-2 depth."""
    from mtlora_amd import mtl_harness as H
    c = golden("c1_model.pt")
    tasks = ["semseg"]
    model = H.build_config_model("c1", seed=0, drop_path_rate=0.2)
    assert [n for n, _ in model.state_dict().items()] == c["state_names"]
    assert [n for n, p in model.named_parameters() if p.requires_grad] == c["trainable_names"]
    assert sum(p.numel() for p in model.parameters()) == c["n_params"] == 28370271
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == c["n_trainable"] == 2451999
    O.det_fill_(list(model.named_parameters()) + list(model.named_buffers()))
    model = model.to(dev()).eval()
    img, tg = O.synthetic_batch(2, 224, tasks, seed=1234)
    img, tg = img.to(dev()), {t: v.to(dev()) for t, v in tg.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        out = model(img)["semseg"]
    loss = H.task_loss("semseg", out, tg["semseg"])
    loss.backward()
    tol = 1e-2 if amp else 1e-3
    assert abs(loss.item() - c["loss"]) <= tol * abs(c["loss"]), (loss.item(), c["loss"])
    cs = c["out"]
    f = out.detach().double().flatten().cpu()
    assert tuple(out.shape) == tuple(cs["shape"])
    assert abs(f.sum().item() - cs["sum"]) <= tol * cs["abssum"]
    assert abs(f.abs().sum().item() - cs["abssum"]) <= tol * cs["abssum"]
    ref = cs["samples"]
    assert (f[cs["idx"]] - ref).abs().max().item() <= tol * max(ref.abs().max().item(), f.abs().max().item())
    named = dict(model.named_parameters())
    eager, worst = None, []
    if amp:  # calibration: the reference's own eager bf16-autocast dataflow on the same parameters, same 16 samples
        cfg = O.swin_t_cfg(224, tasks, 4, 4, drop_path_rate=0.2)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        _, _, eager = _oracle_run(sd, set(c["grads"]), img, tg, cfg, tasks, torch.float32, True, False)
    for n, g in c["grads"].items():
        got = named[n].grad.detach().double().flatten().cpu()
        assert tuple(named[n].shape) == tuple(g["shape"]), n
        scale = max(g["samples"].abs().max().item(), g["abssum"] / got.numel())
        err = (got[g["idx"]] - g["samples"]).abs().max().item() / scale
        if not amp:
            assert err <= 2e-3, (n, err)
            assert abs(got.abs().sum().item() - g["abssum"]) <= 2e-3 * g["abssum"], n
        else:
            e_err = (eager[n].double().flatten()[g["idx"]] - g["samples"]).abs().max().item() / scale
            worst.append((err / max(1e-2, 2.0 * e_err), n, err, e_err))
    if amp:
        worst.sort(reverse=True)
        print("c1 fixture, bf16: worst error / bound", [(round(w[0], 3), w[1]) for w in worst[:4]])
        # two bf16 rounding realizations of the same ill-conditioned element (layers.0.blocks.1.attn.proj.lora_tasks_B: the eager
        # dataflow itself is 4.4 % off the fp32 fixture there) land at 0.82-1.03 of the 2x bound from box to box and from path to path
        # (measured with and without the implicit task hiddens): every tensor within 2x, ONE may reach 2.5x
        assert worst[0][0] <= 1.25 and (len(worst) < 2 or worst[1][0] <= 1.0), worst[:4]
    none = sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None)
    assert none == c["grad_is_none"]


def test_bias_all_trains_frozen_linears_biases():
    """mark_only_lora_as_trainable(bias='all') (reference lora.py:606-617, MODEL.MTLORA.BIAS): `*.linear.bias` trains while W
    stays frozen.  The fused kernels never form that gradient; the autograd wrapper must (column sum of the summed output
    gradients).  A 2-stage backbone, T=2, fp32 vs the oracle in fp64, every trainable gradient compared."""
    from mtlora_amd.lora import mark_only_lora_as_trainable
    from mtlora_amd.swin_transformer_mtlora import SwinTransformerMTLoRA
    tasks = ["semseg", "normals"]
    cfg = O.swin_t_cfg(img_size=56, tasks=tasks, r_shared=8, r_task=4, depths=(2, 2), num_heads=(3, 6), drop_path_rate=0.0, dropout=0.0)
    bb = SwinTransformerMTLoRA(img_size=56, patch_size=4, in_chans=3, num_classes=0, embed_dim=96, depths=[2, 2], num_heads=[3, 6],
                               window_size=7, drop_path_rate=0.0, tasks=tasks, mtlora=cfg["mtlora"])
    O.det_fill_(bb.named_parameters())
    mark_only_lora_as_trainable(bb, bias="all")
    bb = bb.to(dev()).train()
    biases = [n for n, p in bb.named_parameters() if n.endswith("linear.bias")]
    assert biases and all(dict(bb.named_parameters())[n].requires_grad for n in biases)
    assert not any(p.requires_grad for n, p in bb.named_parameters() if n.endswith("linear.weight"))
    x = O.det_tensor("biasall.x", (2, 3, 56, 56), 1.0).to(dev())
    got = bb(x, return_stages=True)
    sum((s * O.det_tensor(f"ba.g.{i}", s.shape, 1.0).to(dev())).sum() +
        sum((tl[t] * O.det_tensor(f"ba.g.{i}.{t}", s.shape, 1.0).to(dev())).sum() for t in tasks)
        for i, (s, tl) in enumerate(got)).backward()
    trainable = {n for n, p in bb.named_parameters() if p.requires_grad}
    P = {k: v.detach().double().clone().requires_grad_(k in trainable) for k, v in bb.state_dict().items() if v.is_floating_point()}
    ref = O.backbone_stages(P, x.double(), cfg)
    sum((s * O.det_tensor(f"ba.g.{i}", s.shape, 1.0).to(dev()).double()).sum() +
        sum((tl[t] * O.det_tensor(f"ba.g.{i}.{t}", s.shape, 1.0).to(dev()).double()).sum() for t in tasks)
        for i, (s, tl) in enumerate(ref)).backward()
    named = dict(bb.named_parameters())
    for n in sorted(trainable):
        r, g = P[n].grad, named[n].grad
        if r is None:
            assert g is None, n
            continue
        assert g is not None, f"{n} got no gradient"
        err = (g.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-9)
        assert err <= 2e-3, (n, err)


@pytest.mark.parametrize("setup", ["buckets_1mb", "one_bucket_34mb_2cores"])
def test_reducer_on_the_real_model_matches_plain_backward(setup):
    """GradReducer(force=True) on the MTLoRA model itself (RCCL at world size 1, per-task head streams on, bf16 autocast):
    3 train steps must leave loss and every parameter BIT-identical to the run without a reducer -- pack / all-reduce /
    unpack and the stream joins may not change a single gradient bit; and the construction-time broadcast must be a no-op.
    ``one_bucket_34mb_2cores`` is the configuration bench.py runs at N > 1 (ONE bucket of >= 34 MB, launched from the hook
    of the last gradient) with the rank pinned to two host cores -- the per-rank host budget of 8 ranks on a 16-core node."""
    _reducer_case(34.0 if setup == "one_bucket_34mb_2cores" else 1.0, pin=setup == "one_bucket_34mb_2cores")


def _reducer_case(bucket_mb, pin=False):
    import os
    import torch.distributed as dist
    from mtlora_amd import mtl_harness as H
    from mtlora_amd.ddp import GradReducer
    tasks = ["semseg", "normals", "sal", "human_parts"]
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        img, tg = H.synthetic_batch(2, 224, tasks, seed=11, device=dev())
        runs = []
        for use in (False, True):
            torch.manual_seed(7)
            from mtlora_amd import functional as Fn
            Fn._seed_counter = 0  # same dropout seeds in both runs
            Fn.droppath_reset()   # ... and the same DropPath draw history
            model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, seed=3).to(dev()).train()
            crit, opt = H.MultiTaskLoss(tasks), H.build_optimizer(model, lr=1e-3)
            red = GradReducer(model.parameters(), bucket_mb=bucket_mb, force=True, buffers=model.buffers()) if use else None
            if use:
                assert red.active and (len(red.buckets) > 3 if bucket_mb < 2 else len(red.buckets) == 1)
            losses = []
            # (pinned only around the steps: building a model with the intra-op pool of a 256-core box squeezed onto two cores took
            # 2.5 minutes)
            affinity, nthr = os.sched_getaffinity(0), torch.get_num_threads()
            if pin:
                os.sched_setaffinity(0, set(sorted(affinity)[:2]))
                torch.set_num_threads(2)
            try:
                for _ in range(3):
                    l, _ = H.train_step(model, crit, opt, img, tg, reducer=red)
                    losses.append(l.clone())
                torch.cuda.synchronize()
            finally:
                if pin:
                    os.sched_setaffinity(0, affinity)
                    torch.set_num_threads(nthr)
            runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
            if red is not None:
                red.remove()
        for a, b in zip(runs[0][0], runs[1][0]):
            assert torch.equal(a, b), (a.item(), b.item())
        for n in runs[0][1]:
            assert torch.equal(runs[0][1][n], runs[1][1][n]), n
    finally:
        if own_pg:
            dist.destroy_process_group()


_TWO_RANK_WORKER = r'''
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mtlora_amd import functional as Fn
from mtlora_amd import mtl_harness as H
from mtlora_amd.ddp import GradReducer

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)                      # BOTH ranks on cuda:0: RCCL cannot do that, gloo stages device tensors through the host
dist.init_process_group("gloo", init_method="env://")
dev = torch.device("cuda", 0)
tasks = ["semseg", "normals", "sal", "human_parts"]
Fn._FACTOR_MIN_M = 0                          # the factor-gradient side stream also for these small layers
model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, seed=3 + 10 * rank).to(dev)
model.eval()                                  # SURVEY 8e: heads' BatchNorm on running statistics, no dropout / DropPath -> per-sample linear


class Probe:                                  # a loss that is a plain mean over samples: rank-mean of gradients == gradient of the joint batch
    def task_low(self, t, lo, w):
        return (lo.float() * w).sum() / lo.shape[0]

    def combine(self, per):
        return sum(per.values()), per


class Keep:                                   # optimizer stand-in: keeps the (reduced) gradients train_step would hand to AdamW
    def __init__(self, params):
        self.param_groups = [{"params": [p for p in params if p.requires_grad]}]
        self.grads = None

    def step(self):
        self.grads = [None if p.grad is None else p.grad.detach().clone() for p in self.param_groups[0]["params"]]

    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]["params"]:
            p.grad = None


g = torch.Generator().manual_seed(99)
img_all = torch.randn(4, 3, 224, 224, generator=g)
with torch.no_grad():
    lows = model(img_all[:1].to(dev), upsample=False, concurrent=False)
w_all = {t: torch.randn(4, *lows[t].shape[1:], generator=g) for t in tasks}
red = GradReducer(model.parameters(), bucket_mb=34.0, buffers=model.buffers())   # broadcasts rank 0's parameters: ranks were seeded differently
assert red.active and red.world == 2 and len(red.buckets) == 1
opt = Keep(list(model.parameters()))
sl = slice(2 * rank, 2 * rank + 2)
for _ in range(2):                            # second step: static_graph path (the used-set agreed on at step one)
    H.train_step(model, Probe(), opt, img_all[sl].to(dev), {t: w_all[t][sl].to(dev) for t in tasks}, clip_grad=0.0, reducer=red,
                 amp_dtype=None)
torch.cuda.synchronize()
mine = opt.grads
names = [n for n, p in model.named_parameters() if p.requires_grad]
# (i) both ranks hold identical gradients
flat = torch.cat([q.reshape(-1).cpu() for q in mine if q is not None])
both = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "ranks disagree after the all-reduce"
# (iii) the structurally unused factors stay without a gradient on both ranks
none = sorted(n for n, q in zip(names, mine) if q is None)
assert none == ["backbone.layers.3.blocks.1.mlp.fc2.lora_shared_A", "backbone.layers.3.blocks.1.mlp.fc2.lora_shared_B"], none
# (ii) == one process on the concatenated batch (same parameters: the broadcast made them rank 0's)
red.remove()
ref = Keep(list(model.parameters()))
H.train_step(model, Probe(), ref, img_all.to(dev), {t: w_all[t].to(dev) for t in tasks}, clip_grad=0.0, reducer=None, amp_dtype=None)
torch.cuda.synchronize()
gmax = max(float(r.abs().max()) for r in ref.grads if r is not None)
worst, bad = 0.0, []
for n, a, r in zip(names, mine, ref.grads):
    assert (a is None) == (r is None), n
    if r is not None:
        e = float((a.double() - r.double()).abs().max()) / max(float(r.abs().max()), 1e-6 * gmax)
        worst = max(worst, e)
        if e > 1e-3:
            bad.append((n, e))
assert not bad, (len(bad), len(names), bad[:8])
dist.barrier()
dist.destroy_process_group()
print("OK", rank, "worst rel err vs joint batch %.2e" % worst)
'''


def test_two_ranks_on_one_gpu_real_model_gloo(tmp_path):
    """VERDICT r05 item 7 / SURVEY 8e: the REAL 4-task model through ``train_step`` + ``GradReducer`` at world size 2 -- two processes
    on cuda:0 over gloo (RCCL cannot place two ranks on one device), task streams, factor-gradient stream and the one-call blocks ON,
    one 34 MB bucket as bench.py uses: (i) both ranks end with identical gradients, (ii) they equal the single-process gradients on the
    concatenated batch at 1e-3 (fp32; heads' BatchNorm on running statistics and a per-sample-mean probe loss, so that the joint
    gradient IS the rank mean), (iii) ``layers.3.blocks.1.mlp.fc2.lora_shared_*`` keep ``grad is None`` on both; ranks are seeded
    differently and made identical by the reducer's broadcast."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_rank_worker.py"
    script.write_text(_TWO_RANK_WORKER)
    port = 29600 + (os.getpid() % 300)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o[-3000:]


def test_factor_gradient_stream_is_bit_identical():
    """train_step runs the MTLoRALinear factor gradients (k_tn) on a second stream next to the backward chain
    (functional.set_factor_stream; joined before clip / AdamW): three steps must leave loss and every parameter bit-identical
    to the single-stream run (same kernels, only their placement changes), with dropout / DropPath on."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    tasks = ["semseg", "normals", "sal", "human_parts"]
    img, tg = H.synthetic_batch(2, 224, tasks, seed=13, device=dev())
    runs = []
    keep, keep_m = H._FACTOR_STREAM, Fn._FACTOR_MIN_M
    try:
        Fn._FACTOR_MIN_M = 0  # (the row threshold would keep these small layers on the main stream)
        for on in (False, True):
            H._FACTOR_STREAM = on
            torch.manual_seed(5)
            Fn._seed_counter = 0
            Fn.droppath_reset()
            model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, seed=3).to(dev()).train()
            crit, opt = H.MultiTaskLoss(tasks), H.build_optimizer(model, lr=1e-3)
            losses = [H.train_step(model, crit, opt, img, tg)[0].clone() for _ in range(3)]
            torch.cuda.synchronize()
            runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
    finally:
        H._FACTOR_STREAM, Fn._FACTOR_MIN_M = keep, keep_m
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b), (a.item(), b.item())
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n


@pytest.mark.parametrize("mode", ["bf16_side_stream", "bf16_one_stream", "fp32"])
def test_block_call_matches_per_layer_calls(mode):
    """the tasks-free blocks as ONE library call per block and direction inside ONE autograd node per run (functional.SwinBlockRunFn,
    mtlora_block_fwd / _bwd): three train steps -- dropout, DropPath, factor packer, clip, AdamW -- must leave the losses and every
    parameter bit-identical to the run through the per-layer autograd Functions (the block call issues the same launches in the same
    order).  depths (3, 2, 3, 2): runs of two blocks (first one applies its own norm1, second one gets it handed over) and of one."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    from mtlora_amd import swin_transformer_mtlora as S
    tasks = ["semseg", "normals", "sal", "human_parts"]
    img, tg = H.synthetic_batch(2, 224, tasks, seed=13, device=dev())
    amp = None if mode == "fp32" else torch.bfloat16
    runs, used = [], []
    keep, keep_m, keep_f = H._FACTOR_STREAM, Fn._FACTOR_MIN_M, S._FUSED_BLOCKS
    calls = [0]
    orig = Fn.SwinBlockRunFn.forward

    def counting(ctx, cl, *a):
        calls[0] += len(cl)
        return orig(ctx, cl, *a)

    Fn.SwinBlockRunFn.forward = staticmethod(counting)
    try:
        Fn._FACTOR_MIN_M = 0
        H._FACTOR_STREAM = mode == "bf16_side_stream"
        for fused in (False, True):
            S.set_fused_blocks(fused)
            calls[0] = 0
            torch.manual_seed(5)
            Fn._seed_counter = 0
            Fn.droppath_reset()
            model = H.build_model(img_size=224, tasks=tasks, depths=(3, 2, 3, 2), r_shared=16, r_task=4, seed=3).to(dev()).train()
            crit, opt = H.MultiTaskLoss(tasks), H.build_optimizer(model, lr=1e-3)
            losses = [H.train_step(model, crit, opt, img, tg, amp_dtype=amp)[0].clone() for _ in range(3)]
            torch.cuda.synchronize()
            used.append(calls[0])
            runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
    finally:
        Fn.SwinBlockRunFn.forward = staticmethod(orig)
        H._FACTOR_STREAM, Fn._FACTOR_MIN_M = keep, keep_m
        S.set_fused_blocks(keep_f)
    assert used == [0, 3 * 6], used  # 2 + 1 + 2 + 1 tasks-free blocks per forward, three steps
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.isfinite(a) and torch.equal(a, b), (a.item(), b.item())
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n


def test_block_call_falls_back_for_hooked_or_merged_blocks():
    """a forward hook inside a block, a merged layer or the reference's window layout keep that block on the per-layer path (the hook
    still fires); eval-mode forwards with gradients take the one-call path and agree with the per-layer path bit for bit."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    from mtlora_amd import swin_transformer_mtlora as S
    tasks = ["semseg", "normals"]
    model = H.build_model(img_size=224, tasks=tasks, depths=(3, 2), num_heads=(3, 6), r_shared=8, r_task=4, seed=3).to(dev()).eval()
    bb = model.backbone
    x = torch.randn(2, 3, 224, 224, device=dev())
    n_calls = [0]
    orig = Fn.SwinBlockRunFn.forward

    def counting(ctx, cl, *a):
        n_calls[0] += len(cl)
        return orig(ctx, cl, *a)

    def run():
        bb.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            st = bb(x, return_stages=True)
        sum((s.float() ** 2).sum() + sum((v.float() ** 2).sum() for v in tl.values()) for s, tl in st).backward()
        return ([s.detach().clone() for s, _ in st],
                {n: p.grad.detach().clone() for n, p in bb.named_parameters() if p.grad is not None})

    Fn.SwinBlockRunFn.forward = staticmethod(counting)
    try:
        prev = S.set_fused_blocks(False)
        ref = run()
        S.set_fused_blocks(True)
        got = run()
        assert n_calls[0] == 3  # stage 0: two blocks, stage 1: one
        for a, b in zip(ref[0], got[0]):
            assert torch.equal(a, b)
        assert ref[1].keys() == got[1].keys()
        for n in ref[1]:
            assert torch.equal(ref[1][n], got[1][n]), n
        # a hook on the second block's fc1: the run ends in front of that block, the hook fires, results unchanged
        seen = []
        h = bb.layers[0].blocks[1].mlp.fc1.register_forward_hook(lambda m, i, o: seen.append(1))
        n_calls[0] = 0
        got = run()
        assert n_calls[0] == 2 and seen
        for n in ref[1]:
            assert torch.equal(ref[1][n], got[1][n]), n
        h.remove()
        # an un-frozen pretrained weight (MTLORA.FREEZE_PRETRAINED False) needs the per-layer Function's dense gradients -- seen per call
        w = bb.layers[1].blocks[0].attn.proj.linear.weight
        w.requires_grad_(True)
        n_calls[0] = 0
        run()
        assert n_calls[0] == 2 and w.grad is not None  # stage 0's run stays fused, stage 1's block does not
        w.requires_grad_(False)
        w.grad = None
        # merged weights (inference): no fused call for that block; with torch.no_grad() none at all
        bb.layers[0].blocks[0].attn.qkv.merge()
        n_calls[0] = 0
        with torch.autocast("cuda", dtype=torch.bfloat16):
            bb(x, return_stages=True)
        assert n_calls[0] == 1  # only stage 1's block
        bb.layers[0].blocks[0].attn.qkv.unmerge()
        n_calls[0] = 0
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            bb(x, return_stages=True)
        assert n_calls[0] == 0
    finally:
        Fn.SwinBlockRunFn.forward = staticmethod(orig)
        S.set_fused_blocks(prev)


def test_block_call_sees_conversions_and_attribute_changes():
    """what the one-call path caches per stage is STRUCTURE only.  After the first grad-enabled forward (a) switching a block
    to the reference's window layout, (b) setting a dropout probability inside a block, (c) ``model.to(torch.bfloat16)`` (LayerNorm and
    factor parameters no longer fp32: their raw pointers must not reach kernels that read fp32) each keep the affected blocks off the
    one-call path -- and the results still agree with the per-layer path; a second stage with another window size but the same
    (B, H, W, C, hidden) gets its own scratch size (the byte-size cache is keyed by the attention geometry too)."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    from mtlora_amd import swin_transformer_mtlora as S
    tasks = ["semseg", "normals"]
    model = H.build_model(img_size=224, tasks=tasks, depths=(3, 2), num_heads=(3, 6), r_shared=8, r_task=4, seed=3).to(dev()).eval()
    bb = model.backbone
    x = torch.randn(2, 3, 224, 224, device=dev())
    n_calls = [0]
    orig = Fn.SwinBlockRunFn.forward

    def counting(ctx, cl, *a):
        n_calls[0] += len(cl)
        return orig(ctx, cl, *a)

    def run(amp=True):
        bb.zero_grad(set_to_none=True)
        n_calls[0] = 0
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            st = bb(x if amp else x.to(torch.bfloat16), return_stages=True)
        sum((s.float() ** 2).sum() + sum((v.float() ** 2).sum() for v in tl.values()) for s, tl in st).backward()
        return [s.detach().float().clone() for s, _ in st]

    Fn.SwinBlockRunFn.forward = staticmethod(counting)
    prev = S.set_fused_blocks(True)
    try:
        ref = run()
        assert n_calls[0] == 3
        blk = bb.layers[0].blocks[0]
        blk.attention_layout = "windows"          # (a) no invalidate_fused_cache() call: seen per call
        got = run()
        assert n_calls[0] == 1                    # stage 0's run ends in front of its first block; stage 1's block stays fused
        for a, b in zip(ref, got):
            assert (b - a).abs().max().item() <= 1e-2 * a.abs().max().item(), "windows layout after the first fused forward"
        blk.attention_layout = "image"
        bb.layers[0].blocks[1].mlp.drop.p = 0.5   # (b) eval mode: still the identity, but the block is no longer the stock one
        got = run()
        assert n_calls[0] == 2
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        bb.layers[0].blocks[1].mlp.drop.p = 0.0
        run()
        assert n_calls[0] == 3
        model.to(torch.bfloat16)                  # (c) every parameter bf16 now, the cached weight copies are gone too
        got = run(amp=False)
        assert n_calls[0] == 0
        for a, b in zip(ref, got):
            assert (b - a).abs().max().item() <= 3e-2 * a.abs().max().item(), "bf16 parameters after the first fused forward"
    finally:
        Fn.SwinBlockRunFn.forward = staticmethod(orig)
        S.set_fused_blocks(prev)
    # same (B, H, W, C, hidden), window 7 vs 8: two entries in the byte-size cache, both calls run
    for ws in (7, 8):
        torch.manual_seed(0)
        lay = S.BasicLayer(dim=96, input_resolution=(56, 56), depth=2, num_heads=3, window_size=ws, tasks=None,
                           mtlora=H.mtlora_namespace([], 8, 4, n_stages=1), layer_idx=0).to(dev())
        for n, q in lay.named_parameters():
            if "lora_" not in n and "norm" not in n:
                q.requires_grad_(False)
        xx = torch.randn(2, 56 * 56, 96, device=dev(), requires_grad=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, _ = lay(xx)
        y.float().pow(2).sum().backward()
        assert torch.isfinite(xx.grad).all()
    # (key layout of functional._block_bytes: (B, H, W, C, hidden, has_norm1, dtype, x_dtype, window_size, ...); counted by content, not by
    # how many entries earlier tests of the process left behind)
    seen = {k[8] for k in Fn._blk_bytes_cache if k[:5] == (2, 56, 56, 96, 384)}
    assert {7, 8} <= seen, seen


def test_factor_packer_one_launch_per_step_is_bit_identical():
    """lora.FactorPacker: the low-rank factors of every MTLoRALinear packed by ONE launch per step (mtlora_linear_pack_table, from the
    second step on: the first records each layer's call signature) instead of one k_pack per layer and forward -- four train steps must
    leave loss and every parameter bit-identical to the run that packs inside every forward call, the library's launch profile must
    show exactly one packing launch per step, and a parameter edited behind the packer's back must fall back to per-call packing."""
    import ctypes
    from mtlora_amd import _lib as L
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    from mtlora_amd.lora import MTLoRALinear
    tasks = ["semseg", "normals", "sal", "human_parts"]
    img, tg = H.synthetic_batch(2, 224, tasks, seed=13, device=dev())
    runs, packs = [], []
    keep = H._PACK_ONCE
    try:
        for on in (False, True):
            H._PACK_ONCE = on
            torch.manual_seed(5)
            Fn._seed_counter = 0
            Fn.droppath_reset()
            model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, seed=3).to(dev()).train()
            crit, opt = H.MultiTaskLoss(tasks), H.build_optimizer(model, lr=1e-3)
            losses = [H.train_step(model, crit, opt, img, tg)[0].clone() for _ in range(3)]
            torch.cuda.synchronize()
            L.check(L.lib().mtlora_prof_begin(100000), "prof_begin")
            losses.append(H.train_step(model, crit, opt, img, tg)[0].clone())
            torch.cuda.synchronize()
            s = L.ProfSummary()
            L.check(L.lib().mtlora_prof_end(ctypes.byref(s)), "prof_end")
            names = {L.lib().mtlora_prof_kind_name(k).decode(): s.count[k] for k in range(L.PROF_KINDS) if s.count[k]}
            packs.append(names.get("k_pack", 0))
            runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
            if on:  # a hand edit (no version bump through .data): the packer cannot see it, so callers must invalidate -- but an
                # in-place op that DOES bump the version makes the layer fall back on its own
                lin = next(m for m in model.modules() if isinstance(m, MTLoRALinear) and m.r > 0)
                assert lin._packed is not None and lin._packed_sig is not None
                with torch.no_grad():
                    lin.lora_shared_B.mul_(1.0)
                x = torch.randn(4, 49, lin.linear.in_features, device=dev())
                sig_before = lin._packed_sig
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    lin(x)
                assert lin._packed_sig == sig_before and lin._packed_sig[1] != tuple(q._version for q in lin._factor_params())
    finally:
        H._PACK_ONCE = keep
    n_layers = sum(1 for m in model.modules() if isinstance(m, MTLoRALinear) and m.r > 0)
    assert packs[0] == n_layers and packs[1] == 1, (packs, n_layers)
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b), (a.item(), b.item())
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n


@pytest.mark.gpu
def test_graphed_train_step_replays_the_eager_step():
    """The whole train step (autocast forward, fused losses, backward with the per-task and factor-gradient side streams, clip,
    AdamW) captured as ONE HIP graph: the capture must validate (three replays from identical state are bit-equal -- no ATen
    reduction with a captured memset is left in the step, tools/find_memsets.py) and three replays must leave the same loss and
    parameters as three eager runs of the same step object.  DropPath off: its masks come from torch's generator, whose
    graph-safe offsets differ from the eager draw; the MTLoRALinear dropout (device seed offset) stays on."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    tasks = ["semseg", "normals", "sal", "human_parts"]
    img, tg = H.synthetic_batch(2, 224, tasks, seed=17, device=dev())
    out = []
    for use_graph in (True, False):
        torch.manual_seed(9)
        Fn._seed_counter = 0
        model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, drop_path_rate=0.0,
                              seed=4).to(dev()).train()
        crit, opt = H.MultiTaskLoss(tasks), H.build_optimizer(model, lr=1e-3, capturable=True)
        try:
            gs = H.GraphedTrainStep(model, crit, opt, img, tg, clip_grad=5.0, warmup=2)
            assert gs.graphed, gs.why
            if not use_graph:
                gs.graphed = False  # the same object's eager path (same seed-offset walk)
            start = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
            losses = [gs().clone() for _ in range(3)]
            torch.cuda.synchronize()
            out.append((losses, {n: (p.detach() - start[n]).double() for n, p in model.named_parameters() if n in start}))
        finally:
            Fn.set_seed_offset(None)
    # same kernels; hipBLASLt may pick another algorithm for the heads' GEMMs while capturing, so not bit-equal: the losses agree
    # to bf16 rounding and the three AdamW updates point the same way
    for a, b in zip(out[0][0], out[1][0]):
        assert abs(a.item() - b.item()) <= 2e-3 * abs(b.item()), (a.item(), b.item())
    num = sum((out[0][1][n] * out[1][1][n]).sum().item() for n in out[0][1])
    den = (sum((out[0][1][n] ** 2).sum().item() for n in out[0][1]) * sum((out[1][1][n] ** 2).sum().item() for n in out[1][1])) ** 0.5
    assert num / den >= 0.98, num / den


@pytest.mark.gpu
def test_fp16_autocast_train_steps_with_grad_scaler():
    """The reference's default mixed precision (torch.cuda.amp.autocast() = fp16 + GradScaler, main.py:329-354) on the HIP path:
    MTLoRALinear / window attention run their fp16 kernels (v_mfma_f32_32x32x16_f16) and the block glue (LayerNorm family,
    residual + DropPath, the heads' BatchNorm + ReLU) its fp16 instantiations -- no ATen LayerNorm / BatchNorm kernel in the step.
    Three scaled steps: finite loss and gradients, the scaler never skips, and the first loss agrees with the bf16 run of the same
    model to mixed-precision accuracy."""
    from mtlora_amd import functional as Fn
    from mtlora_amd import mtl_harness as H
    tasks = ["semseg", "normals", "sal", "human_parts"]
    img, tg = H.synthetic_batch(2, 224, tasks, seed=23, device=dev())
    first = {}
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(11)
        Fn._seed_counter = 0
        Fn.droppath_reset()
        model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, drop_path_rate=0.0, seed=6,
                              DROPOUT=[0.0] * 4).to(dev()).train()
        crit, opt = H.MultiTaskLoss(tasks), H.build_optimizer(model, lr=1e-3)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, enabled=dt == torch.float16)
        for step in range(3):
            prof = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) if step == 2 else None
            if prof is not None:
                prof.__enter__()
            with torch.autocast("cuda", dtype=dt):
                loss, _ = crit.forward_low(model(img, upsample=False), tg)
            assert torch.isfinite(loss).all()
            if step == 0:
                first[dt] = loss.item()
            scaler.scale(loss).backward()
            if prof is not None:
                prof.__exit__(None, None, None)
                ops = {e.key for e in prof.key_averages()}
                stray = {o for o in ops if "layer_norm" in o or "batch_norm" in o}
                assert not stray, (dt, sorted(stray))  # the fused glue kernels served every normalisation, in fp16 as in bf16
            scaler.unscale_(opt)
            grads = [p.grad for p in model.parameters() if p.grad is not None]
            assert grads and all(torch.isfinite(g).all() for g in grads)
            scale_before = scaler.get_scale() if dt == torch.float16 else None
            scaler.step(opt)
            scaler.update()
            if dt == torch.float16:
                assert scaler.get_scale() >= scale_before  # no inf / nan step was skipped
            opt.zero_grad(set_to_none=True)
    assert abs(first[torch.float16] - first[torch.bfloat16]) <= 2e-2 * abs(first[torch.bfloat16]), first
