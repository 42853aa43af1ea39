"""How long does the factor-gradient side stream run on after the main stream has finished the backward pass?
    python tools/side_tail.py [--config c2]      (events on both streams right after loss.backward(); B = the config's batch)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H
from mtlora_amd import functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
a = ap.parse_args()
row = H.config(a.config); tasks = list(row["tasks"]); B = row["batch"]
dev = torch.device("cuda", 0)
model = H.build_config_model(a.config, seed=0, drop_path_rate=0.2).to(dev).train()
crit = H.MultiTaskLoss(tasks)
opt = H.build_optimizer(model, lr=1e-4)
img, tg = H.synthetic_batch(B, row["img_size"], tasks, seed=1234, device=dev)
for _ in range(4):
    H.train_step(model, crit, opt, img, tg)
torch.cuda.synchronize()
side = H._factor_side_stream(dev)
main = torch.cuda.current_stream(dev)
tails, bwds = [], []
for _ in range(8):
    Fn.droppath_begin_step(dev)
    H._factor_packer(model).refresh()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = crit.combine(model(img, upsample=False, per_task_fn=lambda t, lo: crit.task_low(t, lo, tg[t])))
    Fn.droppath_end_step()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main)
    Fn.set_factor_stream(side)
    loss.backward()
    Fn.set_factor_stream(None)
    e1.record(main)
    e2.record(side)
    main.wait_stream(side)
    Fn.factor_stream_joined()
    torch.cuda.synchronize()
    bwds.append(e0.elapsed_time(e1)); tails.append(e1.elapsed_time(e2))
    opt.zero_grad(set_to_none=True)
print(f"{a.config}: backward on the main stream {sum(bwds[2:]) / len(bwds[2:]):.2f} ms; the side stream finishes {sum(tails[2:]) / len(tails[2:]):+.2f} ms after it "
      f"(per step: {', '.join(f'{t:+.2f}' for t in tails)})")
