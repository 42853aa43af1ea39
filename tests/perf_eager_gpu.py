#!/usr/bin/env python3
"""Eager PyTorch-ROCm comparator: the SAME train step as bench.py, but executed with the
oracle's ATen-op dataflow (== the reference's eager path: 6+4T launches per MTLoRALinear, roll / partition /
materialised scores per block) on one MI355X under bf16 autocast.  Lives under tests/ because it imports the
oracle.  Prints one JSON line.

    python tests/perf_eager_gpu.py --steps 5 --warmup 2 [--batch 32]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mtlora_oracle as O  # noqa: E402

TASKS = ("semseg", "normals", "sal", "human_parts")


def eager_gpu_step_factory(B, img_size, device, seed):
    """the oracle's ATen dataflow (== the reference's eager path) on the GPU under bf16 autocast."""
    cfg = O.swin_t_cfg(img_size, TASKS, 64, 4, drop_path_rate=0.2)
    shapes = {("backbone." + k): v for k, v in O.backbone_param_shapes(cfg).items()}
    shapes.update(O.head_param_shapes(cfg, O.NUM_OUTPUT))
    P = {k: v.to(device) for k, v in O.make_params(shapes).items()}
    train = [v.requires_grad_(True) for k, v in P.items()
             if O.trainable_filter(k) and not k.endswith(("running_mean", "running_var"))]
    opt = torch.optim.AdamW(train, lr=5e-4, weight_decay=0.05, fused=True)
    img, tg = O.synthetic_batch(B, img_size, TASKS, seed=seed)
    img, tg = img.to(device), {k: v.to(device) for k, v in tg.items()}
    rng = torch.Generator().manual_seed(0)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = O.full_model(P, img, cfg, train=True, rng=rng)
            loss, _ = O.multi_task_loss({k: v.float() for k, v in out.items()}, tg, TASKS)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in train if p.grad is not None], 5.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
    return step



if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img", type=int, default=448)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    step = eager_gpu_step_factory(a.batch, a.img, dev, 1234)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "images/sec (train step) eager ATen dataflow", "value": round(a.batch * a.steps / dt, 2),
                      "ms_per_step": round(1e3 * dt / a.steps, 2), "batch": a.batch, "dtype": "bf16 autocast",
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
