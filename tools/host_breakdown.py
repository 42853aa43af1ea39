"""Host time of one train step by autograd Function (forward on the main thread, backward on the autograd thread), tiny batch so the GPU
is idle:  python tools/host_breakdown.py [--config c2] [--batch 2]"""
import argparse, collections, inspect, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H
from mtlora_amd import functional as Fn
import mtlora_amd.swin_transformer_mtlora as SW
import mtlora_amd.window_process as WP

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--batch", type=int, default=2)
a = ap.parse_args()
acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(cls, name):
    f = getattr(cls, name)
    def timed(*args, **kw):
        t0 = time.perf_counter()
        try:
            return f(*args, **kw)
        finally:
            e = acc[(cls.__name__, name)]
            e[0] += 1
            e[1] += time.perf_counter() - t0
    setattr(cls, name, staticmethod(timed))


for mod in (Fn, H, SW, WP):
    for n, c in inspect.getmembers(mod, inspect.isclass):
        if issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function and c.__module__ == mod.__name__:
            wrap(c, "forward"); wrap(c, "backward")
row = H.config(a.config); tasks = list(row["tasks"])
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(tasks)
opt = H.build_optimizer(model, lr=1e-4)
img, tg = H.synthetic_batch(a.batch, row["img_size"], tasks, seed=1234, device=dev)
step = lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16)
for _ in range(5):
    step()
torch.cuda.synchronize()
acc.clear()
N = 10
t0 = time.perf_counter()
marks = []
for _ in range(N):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print(f"{a.config} B={a.batch}: {1e3 * wall:.2f} ms per step (host-bound)")
tot = 0.0
for (c, n), (k, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    tot += t
    print(f"  {c + '.' + n:42s} {k / N:6.1f} calls  {1e3 * t / N:7.3f} ms  {1e6 * t / max(k, 1):7.1f} us each")
print(f"  inside Functions: {1e3 * tot / N:.2f} ms of {1e3 * wall:.2f}")
