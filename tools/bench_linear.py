#!/usr/bin/env python3
"""Per-layer kernel time of the MTLoRALinear forward / backward on the C2 shapes (B=32): default path (wave-streaming kernels
where eligible) vs the tiled kernels only (functional.set_tuning(stream=1) / MTLORA_SP=0 at import).
    python tools/bench_linear.py [--shapes s0.qkv ...] [--iters 5] [--kinds] [--knt-only: default path only]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd.lora import MTLoRALinear

TASKS = ["semseg", "normals", "sal", "human_parts"]
SHAPES = {  # name: (M, K, N, with_tasks, x_tasks)
    "s0.qkv": (401408, 96, 288, False, False), "s0.proj": (401408, 96, 96, False, False),
    "s0.fc1": (401408, 96, 384, False, False), "s0.fc2": (401408, 384, 96, False, False),
    "s0.projT": (401408, 96, 96, True, False), "s0.fc1T": (401408, 96, 384, True, True), "s0.fc2T": (401408, 384, 96, True, True),
    "s1.qkv": (100352, 192, 576, False, False), "s1.fc1": (100352, 192, 768, False, False), "s1.fc2": (100352, 768, 192, False, False),
    "s1.fc1T": (100352, 192, 768, True, True), "s1.fc2T": (100352, 768, 192, True, True),
    "s2.qkv": (25088, 384, 1152, False, False), "s2.fc1": (25088, 384, 1536, False, False), "s2.fc2": (25088, 1536, 384, False, False),
    "s3.qkv": (6272, 768, 2304, False, False), "s3.fc1": (6272, 768, 3072, False, False), "s3.fc2": (6272, 3072, 768, False, False),
    "head0": (100352, 272, 1080, None, False), "head1": (100352, 1080, 272, None, False), "merge1": (125440, 768, 384, None, False),
    # c4 (Swin-B/448, B = 16, r = 128 shared and per task): --rs 128 --rt 128
    "b0.qkv": (200704, 128, 384, False, False), "b0.fc1": (200704, 128, 512, False, False), "b0.fc2": (200704, 512, 128, False, False),
    "b1.qkv": (50176, 256, 768, False, False), "b1.fc1": (50176, 256, 1024, False, False), "b1.fc2": (50176, 1024, 256, False, False),
    "b2.qkv": (12544, 512, 1536, False, False), "b2.proj": (12544, 512, 512, False, False),
    "b2.fc1": (12544, 512, 2048, False, False), "b2.fc2": (12544, 2048, 512, False, False),
    "b2.fc1T": (12544, 512, 2048, True, True), "b2.fc2T": (12544, 2048, 512, True, True),
    "b3.qkv": (3136, 1024, 3072, False, False), "b3.fc1": (3136, 1024, 4096, False, False), "b3.fc2": (3136, 4096, 1024, False, False),
}

KINDS = KNT_ONLY = False
RS, RT = 64, 4


def run(name, iters):
    r_s, r_t = RS, RT
    M, K, N, wt, xt = SHAPES[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    if wt is None:
        m = MTLoRALinear(K, N, r=0).to(dev)
    else:
        r = {"shared": r_s, **({t: r_t for t in TASKS} if wt else {})}
        m = MTLoRALinear(K, N, r=r, lora_shared_scale=4.0, lora_task_scale={t: 4.0 for t in TASKS}, lora_dropout=0.05,
                         tasks=TASKS if wt else None).to(dev)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "_B" in n:
                    p.normal_(0, 0.02)
    m.linear.weight.requires_grad_(False)
    if m.linear.bias is not None:
        m.linear.bias.requires_grad_(False)
    m.train()
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, requires_grad=True)
    xts = {t: torch.randn(M, K, device=dev, dtype=torch.bfloat16, requires_grad=True) for t in TASKS} if xt else None
    res = {}
    for mode, env in ((("k_nt", "0"),) if not KNT_ONLY else ()) + (("panel", None),):  # "k_nt": tiled only; "panel": default path
        from mtlora_amd import functional as Fn
        Fn.set_tuning(stream=0 if env is None else 1)  # the switches travel in the descriptor (the library reads no environment)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, yt = m(x, xts)
        outs = [y] + ([yt[t] for t in TASKS] if yt else [])
        gs = [torch.randn_like(o) for o in outs]
        def fwd():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y, yt = m(x, xts)
            return [y] + ([yt[t] for t in TASKS] if yt else [])
        for _ in range(3):
            o = fwd(); torch.autograd.backward(o, gs)
        torch.cuda.synchronize()
        # kernel time only (the library's HIP-event brackets), forward and backward separately: at these sizes one Python
        # call costs more host time than some of the launches take
        import ctypes
        from mtlora_amd import _lib as L
        lib = L.lib()
        def kernel_us(fn):
            torch.cuda.synchronize()
            L.check(lib.mtlora_prof_begin(100000), "prof_begin")
            fn()
            torch.cuda.synchronize()
            sm = L.ProfSummary()
            L.check(lib.mtlora_prof_end(ctypes.byref(sm)), "prof_end")
            if KINDS:
                print("      " + "  ".join(f"{lib.mtlora_prof_kind_name(k).decode()} {1e3 * sm.ms[k] / iters:.1f}us/{sm.count[k] // iters}"
                                           for k in range(L.PROF_KINDS) if sm.count[k]), flush=True)
            return 1e3 * sum(sm.ms[k] for k in range(L.PROF_KINDS)) / iters
        outs_keep = []
        def many_fwd():
            for _ in range(iters):
                outs_keep.append(fwd())
        def many_bwd():
            for o in outs_keep:
                torch.autograd.backward(o, gs)
        tf = kernel_us(many_fwd)
        tb = kernel_us(many_bwd)
        outs_keep.clear()
        res[mode] = (tf, tb)
    es = 2
    T = len(TASKS) if wt else 0
    fb = es * M * ((1 + (T if xt else 0)) * K + (1 + T) * N)
    bb = es * M * ((1 + T) * N + 2 * (1 + (T if xt else 0)) * K)
    res.setdefault("k_nt", res["panel"])
    print(f"{name:9s} M{M} K{K} N{N} T{T}: fwd {res['k_nt'][0]:7.1f} -> {res['panel'][0]:7.1f} us ({fb / res['panel'][0] / 1e6:5.2f} TB/s) | "
          f"bwd {res['k_nt'][1]:7.1f} -> {res['panel'][1]:7.1f} us ({bb / res['panel'][1] / 1e6:5.2f} TB/s)", flush=True)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=list(SHAPES))
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kinds", action="store_true", help="per-kind kernel time of every leg")
    ap.add_argument("--knt-only", action="store_true", help="default path only (no tiled-only leg)")
    ap.add_argument("--rs", type=int, default=64)
    ap.add_argument("--rt", type=int, default=4)
    a = ap.parse_args()
    KINDS, KNT_ONLY, RS, RT = a.kinds, a.knt_only, a.rs, a.rt
    for s in (a.shapes if a.shapes != list(SHAPES) else [n for n in SHAPES if not n.startswith("b")]):
        run(s, a.iters)
