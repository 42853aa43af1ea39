import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from mtlora_amd import mtl_harness as H
from oracle import mtlora_oracle as O
tasks = ["semseg", "normals", "sal", "human_parts"]
dev = torch.device("cuda", 0)
model = H.build_model(img_size=224, tasks=tasks, depths=(2, 2, 2, 2), r_shared=16, r_task=4, drop_path_rate=0.0, seed=3, DROPOUT=[0.0] * 4).to(dev)
model.train()
crit = H.MultiTaskLoss(tasks)
img, tg = H.synthetic_batch(2, 224, tasks, seed=5, device=dev)
loss, per = crit.forward_low(model(img, upsample=False), tg)
loss.backward()
cfg = O.swin_t_cfg(img_size=224, tasks=tasks, r_shared=16, r_task=4, depths=(2, 2, 2, 2), drop_path_rate=0.0, dropout=0.0)
P = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
trainable = {n for n, p in model.named_parameters() if p.requires_grad}
for k in P:
    if k in trainable: P[k].requires_grad_(True)
out = O.full_model(P, img.double().cpu(), cfg, train=True, rng=torch.Generator().manual_seed(0))
rl, _ = O.multi_task_loss(out, {t: v.double().cpu() for t, v in tg.items()}, tasks)
rl.backward()
named = dict(model.named_parameters())
errs = []
for n in sorted(trainable):
    g, r = named[n].grad, P[n].grad
    if r is None or g is None: continue
    sc = max(r.abs().max().item(), 1e-30)
    errs.append(((g.double().cpu() - r).abs().max().item() / sc, n))
errs.sort(reverse=True)
print(os.environ.get("MTLORA_HEAD_GEMM"), "loss", loss.item(), rl.item())
for e, n in errs[:8]: print("  %.2e  %s" % (e, n))
