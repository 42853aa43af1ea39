"""Fused bilinear upsample + loss (+ gradient) kernels at the c2 head shapes (B 32, 56 x 56 -> 448 x 448):
    python tools/bench_uploss.py [--batch 32] [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import functional as Fn
from mtlora_amd.mtl_harness import MultiTaskLoss
from oracle import mtlora_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--low", type=int, default=56)
ap.add_argument("--scale", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, h, S = a.batch, a.low, a.scale
g = torch.Generator().manual_seed(0)
for task in ("semseg", "human_parts", "normals", "sal"):
    C = O.NUM_OUTPUT[task]
    low = (torch.randn(B, h, h, C, generator=g) * 2).to(dev).bfloat16()
    H = h * S
    if task in ("semseg", "human_parts"):
        lab = torch.randint(0, C, (B, 1, H, H), generator=g).float()
        lab[torch.rand(B, 1, H, H, generator=g) < 0.07] = 255.0
    elif task == "normals":
        lab = torch.nn.functional.normalize(torch.randn(B, 3, H, H, generator=g), dim=1)
    else:
        lab = (torch.rand(B, 1, H, H, generator=g) < 0.3).float()
    lab = lab.to(dev)
    kind = MultiTaskLoss.FUSED_KIND[task]
    for _ in range(3):
        v = Fn.UpsampleLossFn.apply(kind, low, lab, S)
    torch.cuda.synchronize()
    import ctypes
    from mtlora_amd import _lib as L
    lib = L.lib()
    L.check(lib.mtlora_prof_begin(10000), "prof_begin")
    for _ in range(a.reps):
        v = Fn.UpsampleLossFn.apply(kind, low, lab, S)
    torch.cuda.synchronize()
    ps = L.ProfSummary()
    L.check(lib.mtlora_prof_end(ctypes.byref(ps)), "prof_end")
    kern_us = ps.ms[14] / a.reps * 1e3 if hasattr(ps, "ms") else float("nan")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        v = Fn.UpsampleLossFn.apply(kind, low, lab, S)
    e1.record()
    torch.cuda.synchronize()
    print(flush=True, end=""); print(f"{task:12s} C={C:2d}  {e0.elapsed_time(e1) / a.reps * 1e3:8.1f} us per call (label statistics + kernel + partial sum), kernel {kern_us:8.1f} us  loss {v.item():.6f}")
