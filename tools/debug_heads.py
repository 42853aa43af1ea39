"""dev aid: where does the fp32 HIP path leave the oracle in the decoder heads?  (c5:4 model, 224 px, B=2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import mtl_harness as H
from oracle import mtlora_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_models as T
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "c5:4"
row = H.config(name); tasks = list(row["tasks"])
model = H.build_config_model(name, seed=3, img_size=224, drop_path_rate=0.0, DROPOUT=[0.0] * 4).to(dev)
T._condition_normals_heads(model, tasks); model.train()
crit = H.MultiTaskLoss(tasks)
img, tg = H.synthetic_batch(2, 224, tasks, seed=5, device=dev)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
for fused in (True, False):
    model.zero_grad(set_to_none=True)
    if fused:
        loss, per = T._hip_loss(model, crit, img, tg, False, concurrent=False)
    else:
        loss, per = crit(model(img), tg)
    loss.backward()
    cfg = O.swin_t_cfg(img_size=224, tasks=tasks, r_shared=row["r_shared"], r_task=row["r_task"], embed_dim=row["embed_dim"],
                       depths=row["depths"], num_heads=row["num_heads"], drop_path_rate=0.0, dropout=0.0)
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    if fused:
        rl, rper, rg = T._oracle_run(sd, trainable, img, tg, cfg, tasks, torch.float64, False, True)
        el, eper, eg = T._oracle_run(sd, trainable, img, tg, cfg, tasks, torch.float32, False, True)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    errs, _ = T._grad_errors(grads, rg); eerrs, _ = T._grad_errors(eg, rg)
    print("fused loss" if fused else "plain ATen loss", "loss", loss.item(), rl.item(), el.item())
    t = tasks[1] if len(tasks) > 1 else tasks[0]
    for n in sorted(errs):
        if f".{t}." in n and ("decoders" in n or "downsampler" in n):
            print(f"  {n:60s} ours {errs[n]:.2e}  eager32 {eerrs[n]:.2e}")
