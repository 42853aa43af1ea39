"""GPU time of the phases of one train step (events on the main stream, whole-phase loops so nothing is traced):
backbone forward / heads + losses forward / backward / clip + AdamW.    python tools/phase_times.py [--config c2] [--batch 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H
from mtlora_amd import functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
row = H.config(a.config); tasks = list(row["tasks"]); B = a.batch or row["batch"]
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(tasks)
opt = H.build_optimizer(model, lr=1e-4)
img, tg = H.synthetic_batch(B, row["img_size"], tasks, seed=1234, device=dev)
full = lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16)


def backbone():
    Fn.droppath_begin_step(dev); H._factor_packer(model).refresh()
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return model.backbone(img, return_stages=True)
    finally:
        Fn.droppath_end_step()


def fwd():
    Fn.droppath_begin_step(dev); H._factor_packer(model).refresh()
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return crit.combine(model(img, upsample=False, per_task_fn=lambda t, lo: crit.task_low(t, lo, tg[t])))[0]
    finally:
        Fn.droppath_end_step()


def fwd_bwd():
    loss = fwd()
    side = H._factor_side_stream(dev)
    Fn.set_factor_stream(side)
    try:
        loss.backward()
    finally:
        Fn.set_factor_stream(None)
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
    for p in model.parameters():
        p.grad = None


def timed(f, n):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def heads_only():
    """heads + losses forward and backward from DETACHED stage outputs (the backbone is run once, outside the timed region)"""
    main = torch.cuda.current_stream(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats_in = [(x.detach(), {t: v.detach().requires_grad_(True) for t, v in tl.items()}) for x, tl in STAGES]
        streams = model.task_streams(dev)
        per = {}
        for t, st in zip(tasks, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                lo = model.decoders({t: model.downsampler[t]([tl[t] for _, tl in feats_in])}, upsample=False, tasks=[t])[t]
                per[t] = crit.task_low(t, lo, tg[t])
        for st in streams:
            main.wait_stream(st)
        loss = crit.combine(per)[0]
    return loss, feats_in


def heads_fwd():
    heads_only()


def heads_fwd_bwd():
    loss, _ = heads_only()
    loss.backward()
    for p in model.parameters():
        p.grad = None


for _ in range(5):
    full()
with torch.no_grad():
    STAGES = backbone()
STAGES = [(x.detach(), {t: v.detach() for t, v in tl.items()}) for x, tl in STAGES]
thf, thfb = timed(heads_fwd, a.steps), timed(heads_fwd_bwd, a.steps)
print(f"heads + losses alone (4 task streams, detached stage outputs): forward {thf:.2f} ms, forward + backward {thfb:.2f} ms")
tb, tf, tfb, tfull = timed(backbone, a.steps), timed(fwd, a.steps), timed(fwd_bwd, a.steps), timed(full, a.steps)
print(f"{a.config} B={B}: backbone fwd {tb:.2f} ms | + heads & losses fwd {tf - tb:.2f} | backward {tfb - tf:.2f} | clip + AdamW {tfull - tfb:.2f} | step {tfull:.2f} ms")
