#!/usr/bin/env python3
"""Per-shape microbenchmark of the hot-path ops at the BASELINE C2 shapes (Swin-T/448, B=32, bf16):
every MTLoRALinear (stage x {qkv, proj, fc1, fc2} x {plain, task-enabled}) forward and backward, and the window
attention forward / backward per stage.  Reports ms and achieved algorithmic GB/s (SURVEY 8d byte formulas).

    python tests/perf_kernels.py [--batch 32] [--only linear|attn]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mtlora_amd import functional as Fn  # noqa: E402
from mtlora_amd.lora import MTLoRALinear  # noqa: E402

TASKS = ["semseg", "normals", "sal", "human_parts"]


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


def bench_linear(M, K, N, T, use_xt, train=True):
    dev = torch.device("cuda")
    tasks = TASKS[:T] if T else None
    r = {"shared": 64, **{t: 4 for t in (tasks or [])}}
    m = MTLoRALinear(K, N, r=r, lora_shared_scale=4.0, lora_task_scale={t: 4.0 for t in (tasks or [])} if tasks else 1.0,
                     lora_dropout=0.05, tasks=tasks).to(dev)
    with torch.no_grad():
        m.lora_shared_B.normal_(0, 0.02)
        for t in tasks or []:
            m.lora_tasks_B[t].normal_(0, 0.02)
    m.linear.weight.requires_grad_(False)
    m.linear.bias.requires_grad_(False)
    m.train(train)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, requires_grad=True)
    xt = {t: torch.randn(M, K, device=dev, dtype=torch.bfloat16, requires_grad=True) for t in tasks} if (tasks and use_xt) else None
    y, yt = m(x, xt)
    outs = [y] + ([yt[t] for t in tasks] if tasks else [])
    gs = [torch.randn_like(o) for o in outs]
    f_ms = timeit(lambda: m(x, xt))

    def fb():
        y, yt = m(x, xt)
        outs = [y] + ([yt[t] for t in tasks] if tasks else [])
        torch.autograd.backward(outs, gs)
    fb_ms = timeit(fb)
    nx = T if use_xt else 0
    fwd_b = 2 * (M * K * (1 + nx) + M * N * (1 + T))
    bwd_b = 2 * (M * N * (1 + T) + 2 * M * K * (1 + nx))
    b_ms = fb_ms - f_ms
    return f_ms, b_ms, fwd_b / f_ms / 1e6, bwd_b / b_ms / 1e6


def bench_attn(B, H, C, nH, shift):
    dev = torch.device("cuda")
    N = 49
    qkv = torch.randn(B, H, H, 3 * C, device=dev, dtype=torch.bfloat16, requires_grad=True)
    bias = torch.randn(nH, N, N, device=dev, requires_grad=True)
    sys.path.insert(0, ROOT)
    from mtlora_amd.swin_transformer_mtlora import _shift_regions
    mask = None
    mask_t = _shift_regions(H, H, 7, shift).to(torch.int32).to(dev) if shift else None
    meta = Fn.AttnMeta(B=B, H=H, W=H, window_size=7, shift=shift, num_heads=nH, head_dim=32, image_layout=True, scale=32 ** -0.5)
    out = Fn.WindowAttentionFn.apply(meta, qkv, bias, mask, mask_t)
    g = torch.randn_like(out)
    f_ms = timeit(lambda: Fn.WindowAttentionFn.apply(meta, qkv, bias, mask, mask_t))

    def fb():
        o = Fn.WindowAttentionFn.apply(meta, qkv, bias, mask, mask_t)
        o.backward(g)
    fb_ms = timeit(fb)
    M = B * H * H
    return f_ms, fb_ms - f_ms, 4 * M * C * 2 / f_ms / 1e6, 7 * M * C * 2 / (fb_ms - f_ms) / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B = a.batch
    stages = [(96, 112, 3), (192, 56, 6), (384, 28, 12), (768, 14, 24)]
    if a.only in ("", "linear"):
        print(f"{'layer':34s} {'fwd ms':>8s} {'bwd ms':>8s} {'fwd GB/s':>9s} {'bwd GB/s':>9s}")
        tot_f = tot_b = 0.0
        for si, (C, H, nH) in enumerate(stages):
            M = B * H * H
            for name, K, N in (("qkv", C, 3 * C), ("proj", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)):
                for T, xt in ((0, False),) + (((4, name != "proj"),) if name != "qkv" else ()):
                    f, b, fg, bg = bench_linear(M, K, N, T, xt)
                    depth = [2, 2, 6, 2][si]
                    cnt = 1 if T else (depth if name == "qkv" else depth - 1)
                    tot_f += f * cnt
                    tot_b += b * cnt
                    print(f"s{si} {name:5s} M={M:6d} K={K:4d} N={N:4d} T={T} x{cnt} {f:8.3f} {b:8.3f} {fg:9.0f} {bg:9.0f}")
        print(f"model total (weighted by call count): fwd {tot_f:.2f} ms  bwd {tot_b:.2f} ms")
    if a.only in ("", "attn"):
        print(f"{'attention':34s} {'fwd ms':>8s} {'bwd ms':>8s} {'fwd GB/s':>9s} {'bwd GB/s':>9s}")
        for si, (C, H, nH) in enumerate(stages):
            for shift in (0, 3):
                f, b, fg, bg = bench_attn(B, H, C, nH, shift)
                print(f"s{si} attn  H={H:3d} C={C:4d} nH={nH:2d} shift={shift} {f:8.3f} {b:8.3f} {fg:9.0f} {bg:9.0f}")
