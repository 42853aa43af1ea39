"""dev aid: BatchNormReluFn backward inside the c5:4 model (fp32) vs an fp64 evaluation on the SAME dy / x."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H, functional as Fn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_models as T
dev = torch.device("cuda", 0)
name = "c5:4"
row = H.config(name); tasks = list(row["tasks"])
model = H.build_config_model(name, seed=3, img_size=224, drop_path_rate=0.0, DROPOUT=[0.0] * 4).to(dev)
T._condition_normals_heads(model, tasks); model.train()
crit = H.MultiTaskLoss(tasks)
img, tg = H.synthetic_batch(2, 224, tasks, seed=5, device=dev)
orig = Fn.BatchNormReluFn.backward
seen = []
def spy(ctx, dy):
    out = orig(ctx, dy)
    x, saves = ctx.saved_tensors
    seen.append((dy.detach().clone(), x.detach().clone(), saves.detach().clone(), [o.detach().clone() if o is not None else None for o in out[:3]]))
    return out
Fn.BatchNormReluFn.backward = staticmethod(spy)
loss, per = T._hip_loss(model, crit, img, tg, False, concurrent=False)
loss.backward()
for i, (dy, x, saves, (dx, dg, db)) in enumerate(seen[:3]):
    X, G = x.double(), dy.double()
    mean, var = X.mean(0), X.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    print("saved mean err", (saves[0].double() - mean).abs().max().item(), "rstd rel err", ((saves[1].double() - rstd) / rstd).abs().max().item())
    xh = (X - mean) * rstd
    # gamma = scale / rstd
    gamma = saves[2].double() / saves[1].double()
    beta = saves[3].double() + saves[0].double() * saves[2].double()
    y = xh * gamma + beta
    g = torch.where(y > 0, G, torch.zeros_like(G))
    rdb, rdg = g.sum(0), (g * xh).sum(0)
    rdx = gamma * rstd * (g - g.mean(0) - xh * (g * xh).mean(0))
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
    print(i, "R,C", tuple(x.shape), "db", rel(db, rdb), "dg", rel(dg, rdg), "dx", rel(dx, rdx), "| |sum g|/sum|g| min", (rdb.abs() / g.abs().sum(0).clamp_min(1e-30)).min().item(),
          "frac y in (-1e-6,1e-6)", ((y.abs() < 1e-6).double().mean()).item(), "dy dtype", dy.dtype, x.dtype)
