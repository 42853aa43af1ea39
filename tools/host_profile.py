"""cProfile of the host side of one train step (tiny batch: the GPU is idle, so wall = Python / autograd / launch work).
    python tools/host_profile.py [--batch 2]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--config", default="c2")
ap.add_argument("--capturable", action="store_true")
ap.add_argument("--forward", action="store_true", help="profile the forward + loss only (no backward / optimizer)")
ap.add_argument("--top", type=int, default=28)
ap.add_argument("--sort", default="tottime")
a = ap.parse_args()
row = H.config(a.config)
TASKS = tuple(row["tasks"])
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(TASKS)
opt = H.build_optimizer(model, lr=1e-4, capturable=a.capturable)
img, tg = H.synthetic_batch(a.batch, row["img_size"], TASKS, seed=1234, device=dev)
def fwd_only():
    with torch.autocast("cuda", dtype=torch.bfloat16):  # train_step's forward: fused per-task upsample + loss
        return crit.combine(model(img, upsample=False, per_task_fn=lambda t, lo: crit.task_low(t, lo, tg[t])))
step = fwd_only if a.forward else (lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16))
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("ms/step (host-bound at this batch): %.2f" % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats(a.sort).print_stats(a.top)
