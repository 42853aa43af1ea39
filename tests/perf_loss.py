#!/usr/bin/env python3
"""Microbenchmark of the fused upsample+loss kernels at the BASELINE C2 shapes (B=32, 112x112 -> 448x448, bf16)
against the ATen sequence they replace (interpolate + task_loss + backward).   python tests/perf_loss.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mtlora_amd import functional as Fn  # noqa: E402
from mtlora_amd.mtl_harness import MultiTaskLoss, task_loss, NUM_OUTPUT  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


if __name__ == "__main__":
    dev = torch.device("cuda")
    for (B, h, S) in ((32, 56, 8), (32, 112, 4)):  # c2's heads: 56 x 56 maps, x 8; the x 4 geometry for comparison
      print(f"B={B} {h}x{h} -> x{S}")
      for t in ("semseg", "human_parts", "normals", "sal"):
          C = NUM_OUTPUT[t]
          low = torch.randn(B, h, h, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
          if t in ("semseg", "human_parts"):
              lab = torch.randint(0, C, (B, 1, h * S, h * S), device=dev).float()
          elif t == "normals":
              lab = torch.nn.functional.normalize(torch.randn(B, 3, h * S, h * S, device=dev), dim=1)
          else:
              lab = (torch.rand(B, 1, h * S, h * S, device=dev) < 0.3).float()

          def fused():
              low.grad = None
              Fn.UpsampleLossFn.apply(MultiTaskLoss.FUSED_KIND[t], low, lab, S).backward()

          def plain():
              low.grad = None
              task_loss(t, torch.nn.functional.interpolate(low.permute(0, 3, 1, 2), scale_factor=S, mode="bilinear"), lab).backward()

          print(f"{t:12s} C={C:2d}  fused {timeit(fused):7.3f} ms   ATen {timeit(plain):7.3f} ms")
