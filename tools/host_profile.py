"""cProfile of the host side of one train step (tiny batch: the GPU is idle, so wall = Python / autograd / launch work).
    python tools/host_profile.py [--batch 2]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
a = ap.parse_args()
TASKS = ("semseg", "normals", "sal", "human_parts")
dev = torch.device("cuda", 0)
model = H.build_model(img_size=448, tasks=TASKS, r_shared=64, r_task=4, drop_path_rate=0.2, seed=0).to(dev).train()
crit = H.MultiTaskLoss(TASKS)
opt = H.build_optimizer(model, lr=1e-4)
img, tg = H.synthetic_batch(a.batch, 448, TASKS, seed=1234, device=dev)
step = lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16)
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
st.sort_stats("tottime").print_stats(28)
