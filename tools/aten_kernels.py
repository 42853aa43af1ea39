"""Which ATen / hipBLASLt / copy KERNELS does one train step still launch, from which op and which line of this repo, and what do
they cost on the GPU (torch.profiler, full batch):
    python tools/aten_kernels.py [--config c2] [--top 60]
Library kernels (anonymous-namespace k_*) are summed into one line; everything else is listed by (kernel family, op, site)."""
import argparse, collections, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
row = H.config(a.config); tasks = list(row["tasks"]); B = a.batch or row["batch"]
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(tasks)
opt = H.build_optimizer(model, lr=1e-4)
img, tg = H.synthetic_batch(B, row["img_size"], tasks, seed=1234, device=dev)
step = lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()


def family(k):
    if "k_" in k and ("anonymous namespace" in k or "_GLOBAL__N_" in k):
        return None
    if k.startswith("Cijk"):
        return "hipBLASLt " + re.sub(r"_MI.*", "", k)[:40]
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    m = re.search(r"at::native::(\w+)<[^,]*, *(?:at::native::)?([\w:]+)", k)
    if m:
        return f"{m.group(1)}<{m.group(2)[:36]}>"
    return k[:60]


def site(evt):
    for fr in (evt.stack or []):
        if REPO in fr and "tools/" not in fr:
            return fr.replace(REPO + "/", "").split(" ")[0] if ":" in fr else fr
    e = evt.cpu_parent
    while e is not None:
        for fr in (e.stack or []):
            if REPO in fr and "tools/" not in fr:
                return fr.replace(REPO + "/", "")
        e = e.cpu_parent
    return "?"


agg = collections.defaultdict(lambda: [0, 0.0])
lib_n, lib_t, tot_n, tot_t = 0, 0.0, 0, 0.0
for evt in prof.events():
    if not evt.kernels or evt.cpu_children and any(c.kernels for c in evt.cpu_children):
        continue
    for k in evt.kernels:
        fam = family(k.name)
        dur = k.duration if hasattr(k, "duration") else k.device_time
        tot_n += 1; tot_t += dur
        if fam is None:
            lib_n += 1; lib_t += dur
            continue
        key = (fam, evt.name, site(evt))
        agg[key][0] += 1; agg[key][1] += dur
print(f"{a.config} B={B}: {tot_n} kernels, {tot_t / 1e3:.2f} ms of kernel time in one step; library k_*: {lib_n} launches {lib_t / 1e3:.2f} ms; "
      f"other: {tot_n - lib_n} launches {(tot_t - lib_t) / 1e3:.2f} ms")
for (fam, op, st), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"  {n:4d} x {t / n:7.1f} us = {t / 1e3:6.3f} ms  {fam:48s} {op:28s} {st}")
