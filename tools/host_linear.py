"""host cost of ONE MTLoRALinear call (tiny M: the GPU is idle, wall = Python + autograd + ctypes + launch):
    python tools/host_linear.py [--tasks] [--profile]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd.lora import MTLoRALinear, FactorPacker

ap = argparse.ArgumentParser()
ap.add_argument("--tasks", action="store_true")
ap.add_argument("--profile", action="store_true")
ap.add_argument("--n", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda", 0)
TASKS = ["semseg", "normals", "sal", "human_parts"]
r = {"shared": 64, **({t: 4 for t in TASKS} if a.tasks else {})}
m = MTLoRALinear(384, 384, r=r, lora_shared_scale=4.0, lora_task_scale={t: 4.0 for t in TASKS}, lora_dropout=0.05,
                 tasks=TASKS if a.tasks else None).to(dev)
m.linear.weight.requires_grad_(False); m.linear.bias.requires_grad_(False)
m.train()
pk = FactorPacker(m)
x = torch.randn(64, 384, device=dev, dtype=torch.bfloat16, requires_grad=True)


def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, yt = m(x)
    return y if yt is None else y + sum(yt.values())


def step():
    pk.refresh()
    fwd().sum().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.n):
    fwd()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(a.n):
    step()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"forward call {1e6 * (t1 - t0) / a.n:.1f} us;  refresh + forward + sum + backward {1e6 * (t2 - t1) / a.n:.1f} us  (tasks={a.tasks})")
if a.profile:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(a.n):
        fwd()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
