"""The per-task Downsampler + HighResolutionHead + fused upsample/loss chains ALONE (forward + backward from detached stage outputs),
on ONE stream so that rocprofv3 durations are not inflated by overlap:
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -- python tools/heads_profile.py [--config c2] [--iters 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H
from mtlora_amd import functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--streams", action="store_true", help="one stream per task as train_step does (default: everything on one stream)")
a = ap.parse_args()
row = H.config(a.config); tasks = list(row["tasks"]); B = row["batch"]
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(tasks)
img, tg = H.synthetic_batch(B, row["img_size"], tasks, seed=1234, device=dev)
Fn.droppath_begin_step(dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    stages = model.backbone(img, return_stages=True)
Fn.droppath_end_step()
stages = [(x.detach(), {t: v.detach() for t, v in tl.items()}) for x, tl in stages]


def once():
    main = torch.cuda.current_stream(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ins = [{t: v.detach().requires_grad_(True) for t, v in tl.items()} for _, tl in stages]
        per = {}
        for i, t in enumerate(tasks):
            st = model.task_streams(dev)[i] if a.streams else main
            st.wait_stream(main)
            with torch.cuda.stream(st):
                lo = model.decoders({t: model.downsampler[t]([tl[t] for tl in ins])}, upsample=False, tasks=[t])[t]
                per[t] = crit.task_low(t, lo, tg[t])
            main.wait_stream(st)
        loss = crit.combine(per)[0]
    loss.backward()
    for p in model.parameters():
        p.grad = None


for _ in range(3):
    once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    once()
e1.record(); torch.cuda.synchronize()
print(f"heads fwd + bwd, {'task streams' if a.streams else 'one stream'}: {e0.elapsed_time(e1) / a.iters:.2f} ms per step ({a.iters + 3} iterations in the trace)")
