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


for _ in range(5):
    full()
tb, tf, tfb, tfull = timed(backbone, a.steps), timed(fwd, a.steps), timed(fwd_bwd, a.steps), timed(full, a.steps)
print(f"{a.config} B={B}: backbone fwd {tb:.2f} ms | + heads & losses fwd {tf - tb:.2f} | backward {tfb - tf:.2f} | clip + AdamW {tfull - tfb:.2f} | step {tfull:.2f} ms")
