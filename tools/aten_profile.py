"""torch.profiler view of one train step: which ATen ops launch the non-library kernels (run on the GPU box).
    python tools/aten_profile.py [--batch 32] -> table of CPU ops by device time + their kernel counts"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--stacks", type=str, default="")
a = ap.parse_args()
TASKS = ("semseg", "normals", "sal", "human_parts")
dev = torch.device("cuda", 0)
model = H.build_model(img_size=448, tasks=TASKS, r_shared=64, r_task=4, drop_path_rate=0.2, seed=0).to(dev).train()
crit = H.MultiTaskLoss(TASKS)
opt = H.build_optimizer(model, lr=5e-4 * a.batch / 512.0)
img, tg = H.synthetic_batch(a.batch, 448, TASKS, seed=1234, device=dev)
step = lambda: H.train_step(model, crit, opt, img, tg, clip_grad=5.0, amp_dtype=torch.bfloat16)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(a.stacks)) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_device_time_total", row_limit=45, max_name_column_width=70))
if a.stacks:
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_device_time_total", row_limit=60, max_name_column_width=60))
