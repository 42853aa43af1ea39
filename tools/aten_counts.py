"""Which host-side ops launch the kernels of one train step, BY COUNT (launch-bound work shows up here, not in device time):
    python tools/aten_counts.py [--config c2] [--batch 4]  -> kernel launches per step, ATen ops by call count with the Python
    frame that issued them."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
row = H.config(a.config); tasks = list(row["tasks"])
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(tasks)
opt = H.build_optimizer(model, lr=1e-4)
img, tg = H.synthetic_batch(a.batch, row["img_size"], tasks, seed=1234, device=dev)
step = lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("device kernels / memcpys in the step:", len(kern))
names = collections.Counter(e.name[:60] for e in kern)
for n, c in names.most_common(25):
    print(f"  {c:5d}  {n}")
print("--- CPU ops that launch, by count (op | innermost repo frame)")
cnt = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if not e.kernels:
        continue
    fr = next((s for s in (e.stack or []) if "/repo/" in s or "mtlora_amd" in s or "torch/optim" in s or "clip_grad" in s), "?")
    cnt[(e.name, fr.split("/")[-1][:70])] += 1
for (n, fr), c in cnt.most_common(40):
    print(f"  {c:5d}  {n:28s} {fr}")
print("--- device memsets / fills (what a HIP-graph capture of the step would bake in): op | stack")
ms = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    ks = [k.name for k in e.kernels if "Memset" in k.name or "fillBuffer" in k.name]
    if not ks:
        continue
    st = [s.split("/")[-1][:60] for s in (e.stack or []) if "mtlora_amd" in s or "bench.py" in s or "torch/optim" in s or "clip_grad" in s or "nn/functional" in s or "autograd/function" in s or "tools/" in s][:4]
    if not st:
        st = [s.split("/")[-1][:50] for s in (e.stack or [])][:4] or ["(autograd engine thread: no Python stack)"]
    ms[(e.name, " < ".join(st))] += len(ks)
for (n, st), c in ms.most_common(30):
    print(f"  {c:4d}  {n:30s} {st}")
