"""Which ATen calls (and which python lines) of one train step launch kernels whose name contains a pattern?
    python tools/find_kernel_sites.py --kernel FillFunctor [--config c2] [--batch 32]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--kernel", default="FillFunctor")
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    n = sum(1 for k in e.kernels if a.kernel in k.name)
    if not n:
        continue
    site = next((s for s in (e.stack or []) if "mtlora_amd" in s or "bench.py" in s), (e.stack or ["?"])[0] if e.stack else "?")
    cnt[(e.name, str(e.input_shapes)[:80], site.replace(REPO + "/", "")[:110])] += n
print(f"kernels matching {a.kernel!r} in one step at batch {a.batch}: {sum(cnt.values())}")
for (n, sh, site), c in cnt.most_common(40):
    print(f"  {c:4d}  {n:22s} {sh:80s} {site}")
