"""Task-enabled Mlp (fc1 -> GELU -> fc2 with x_tasks) at the c2 / c5 shapes: implicit task hiddens (csrc/hid.hip) against the per-layer path,
kernel time per launch kind from the library's own HIP-event profiler.
    python tools/bench_hid.py [--stage 0] [--tasks 4] [--rank 4] [--reps 5]"""
import argparse, collections, ctypes, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd import _lib as L, functional as Fn, mtl_harness as H
from mtlora_amd.swin_transformer_mtlora import Mlp

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=0)
ap.add_argument("--tasks", type=int, default=4)
ap.add_argument("--rank", type=int, default=4)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--embed", type=int, default=96)
a = ap.parse_args()
dev = torch.device("cuda", 0)
C = a.embed << a.stage
M = a.batch * (112 >> a.stage) ** 2
tasks = [f"t{i}" for i in range(a.tasks)]
torch.manual_seed(0)
ns = H.mtlora_namespace(tasks, r_shared=64, r_task=a.rank, scale=4.0, dropout=0.05)
mlp = Mlp(C, 4 * C, lora=True, tasks=tasks, mtlora=ns, layer_idx=0).to(dev).train()
with torch.no_grad():
    for n_, p_ in mlp.named_parameters():
        if "lora_" in n_:
            p_.normal_(0, 0.05)
        else:
            p_.requires_grad_(False)
xs = [torch.randn(M, C, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(1 + a.tasks)]
gys = [torch.randn(M, C, device=dev, dtype=torch.bfloat16) for _ in range(1 + a.tasks)]
lib = L.lib()


def run():
    for x in xs:
        x.grad = None
    mlp.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, yt = mlp(xs[0], {t: xs[1 + i] for i, t in enumerate(tasks)})
    torch.autograd.backward([y] + [yt[t] for t in tasks], gys)


for hid in (False, True):
    Fn.set_mlp_hid(hid)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    dump = tempfile.mktemp()
    os.environ["MTLORA_PROF_DUMP"] = dump
    L.check(lib.mtlora_prof_begin(100000), "prof_begin")
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    s = L.ProfSummary()
    L.check(lib.mtlora_prof_end(ctypes.byref(s)), "prof_end")
    tot = collections.OrderedDict()
    for line in open(dump):
        p = line.rstrip("\n").split(",")
        if len(p) < 5:
            continue
        key = p[0] + " " + (p[5][:44] if len(p) > 5 else "")
        tot[key] = tot.get(key, 0.0) + float(p[4])
    os.remove(dump)
    total = sum(tot.values()) / a.reps
    print(f"{'implicit task hiddens' if hid else 'per-layer path':24s} M={M} C={C} T={a.tasks} r_t={a.rank}: {total:8.3f} ms of library kernels per fwd+bwd")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"      {k:70s} {v / a.reps:8.3f} ms")
