"""Which lines of THIS repo issue the ATen ops of one train step (forward and backward), by count:
    python tools/aten_sites.py [--config c2] [--batch 2]      (a TorchDispatchMode: every op that reaches the dispatcher is seen)"""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from mtlora_amd import mtl_harness as H

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--views", action="store_true", help="list the view-type ops instead (each one is an autograd node when its input requires grad)")
a = ap.parse_args()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWS = ("view", "reshape", "permute", "transpose", "t.default", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
         "as_strided", "_unsafe_view", "unbind", "split", "chunk", "empty", "_local_scalar", "is_", "size", "stride", "record_stream")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        isview = any(v in name for v in VIEWS) and not any(v in name for v in ("empty", "_local_scalar", "is_", "size", "stride", "record_stream", "detach"))
        if (isview if a.views else not any(v in name for v in VIEWS)):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if fr.filename.startswith(REPO) and "tools/" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, REPO)}:{fr.lineno} {fr.name}"
                    break
            self.n[(name.replace("aten.", ""), site)] += 1
        return func(*args, **(kwargs or {}))


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
with Sites() as m:
    step()
torch.cuda.synchronize()
print(f"{sum(m.n.values())} {'view-type' if a.views else 'non-view'} ATen ops in one step ({a.config}, B = {a.batch})")
for (op, site), k in m.n.most_common(a.top):
    print(f"  {k:4d}  {op:34s} {site}")
