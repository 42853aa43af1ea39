#!/usr/bin/env python3
"""bench.py -- images/sec of one MTLoRA train step (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3 [--config c2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch: bf16-autocast forward of the MTLoRA Swin backbone +
HRNet heads + weighted multi-task loss, backward, [RCCL all-reduce of the trainable gradients], clip_grad_norm_(5.0),
AdamW, zero_grad -- dropout 0.05 and DropPath 0.2 active (train mode).  Per-GPU batch is fixed (weak scaling); inputs
are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

--config selects a row of mtlora_amd.mtl_harness.CONFIGS (the BASELINE.json configs):
    c2 (default)  Swin-T/448, 4 tasks, r_shared 64 / r_task 4, B=32      <- the configuration the metric is quoted on
    c1            Swin-T/224, 1 task, r=4, B=2 (the reference's CPU-runnable plumbing case)
    c4            Swin-B/448, 4 tasks, r=128 shared and per task, B=16   (near machine balance)
    c5:<r>        Swin-T/448, 8 synthetic tasks, r in {4,16,64,256}      (skinny-GEMM HBM regime)

Extra objects in the line (this tier's contract):
  roofline      the dominant HIP kernel (k_nt, the fused MTLoRALinear GEMM), HOT-PATH launches only (the callers' plain
                rank-0 GEMMs have their own kinds): SURVEY 8(d) algorithmic bytes / their HIP-event durations, measured
                over extra profiled steps right after the timed region (events on the launch stream; mtlora_prof_begin).
                `frac` follows 8(d) exactly (no GELU traffic); `launched` is the same with the fused GELU write / gate
                read counted as useful bytes; `mfma` prices the same launches against the dense bf16 MFMA peak;
                `linear_path` covers every kernel of the MTLoRALinear path (k_nt + k_tn + pack / reduce / sum).
  eager_gpu     the north star's ">= 4x" comparator, timed in the same process: the oracle's ATen-op dataflow (== the
                reference's eager PyTorch-ROCm path: 6+4T launches per MTLoRALinear, roll / partition / materialised
                scores per block) on the same GPU, same config, same batch, bf16 autocast, same train step, in a process
                of its own without this file's MIOPEN_FIND_MODE default; 3 warm-up + 10 timed steps, median (min / max reported).
  cpu_baseline  the oracle (a plain-PyTorch port of the reference) on this box's host cores on a bounded sample
                (B=2, fp32) -- kind "port".  Reported baselines, not targets.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if "--eager-leg" not in sys.argv:  # (the eager comparator's own process keeps MIOpen's defaults: see eager_leg)
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")  # no exhaustive conv search for the (few) MIOpen ops left

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s achievable)
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: measured float4 copy
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", help="c1 | c2 (default, the BASELINE metric) | c4 | c5:<r>")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=0, help="steps of the profiled leg (default: min(max(steps, 2), 5); tools/pmc.sh uses 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-gpu", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as captured HIP graph(s) (validated: replays from identical state must agree, else the eager "
                         "step runs and the reason is reported); faster when the host is the limit (c1: 2x), not at c2 -- DESIGN.md 5")
    ap.add_argument("--capturable", action="store_true", help="AdamW(capturable=True) in the eager step too (step counters on the device)")
    ap.add_argument("--force-reducer", action="store_true", help="run the gradient pack / RCCL all-reduce / unpack path even at 1 rank")
    ap.add_argument("--no-fused-loss", action="store_true", help="final upsample + losses through ATen instead of csrc/loss.hip")
    ap.add_argument("--cores", type=int, default=-1,
                    help="pin this rank to N host cores of its own (sched_setaffinity + torch.set_num_threads; rank r takes cores "
                         "[r N, (r + 1) N) of the affinity mask).  Default -1 = 2: the step is issued by two threads (python + autograd), "
                         "and letting them migrate over a 256-core host costs 0.4-2.5 %% (DESIGN.md 7); 0 = leave the affinity alone")
    ap.add_argument("--eager-leg", default="", help=argparse.SUPPRESS)  # child process of the eager_gpu leg: --eager-leg <config> --batch B
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short c4 / c5:4 legs the default (c2, 1 GPU) run appends under `other_configs`")
    return ap.parse_args()


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def time_steps(step_fn, steps, warmup, world):
    for _ in range(warmup):
        step_fn()
    barrier(world)
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        step_fn()
        host += time.perf_counter() - h0   # time the host spends ISSUING a step (no sync inside the step)
    t_issue = time.perf_counter() - t0
    barrier(world)
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, t_issue


def _static_traffic(cfg_name, n_hot, n_all):
    """PMC-measured HBM bytes of the committed profile of this config (separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes cannot be
    collected from inside the timed process): STATIC, quoted only when the launch structure matches the profiled run.  r04 profiles
    attribute the counters per launch kind (tools/pmc_traffic.py: hot-path launches only); older ones cover every launch of the GEMM
    kernels, i.e. the callers' rank-0 GEMMs too -- labelled as such."""
    suffix = "" if cfg_name == "c2" else "_" + cfg_name.replace(":", "_")
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        fn = f"{rnd}_pmc_traffic{suffix}.json"
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        hp = pm.get("hot_path")
        if hp and abs(hp["launches_per_step"] - n_hot) < 0.5:
            lp = pm.get("linear_path", {})
            return {"traffic": round(hp["traffic_bytes_per_launch"]), "traffic_ratio": round(hp["traffic_ratio"], 3),
                    "traffic_source": f"static: profiles/{fn} (hot-path launches only, per-kind attribution)",
                    "linear_traffic_GB": round(lp.get("traffic_bytes_per_step", 0.0) / 1e9, 2) or None,
                    "linear_traffic_ratio": round(lp["traffic_ratio"], 3) if lp.get("traffic_ratio") else None}
        if "k_nt" in pm and abs(pm["k_nt"]["launches_per_step"] - n_all) < 0.5:
            return {"traffic": round(pm["k_nt"]["traffic_bytes_per_launch"]), "traffic_ratio": None,
                    "traffic_source": f"static: profiles/{fn} (ALL launches of the GEMM kernels incl. the callers' rank-0 GEMMs: "
                                      "not comparable with alg_bytes_per_launch)", "linear_traffic_GB": None, "linear_traffic_ratio": None}
    return {"traffic": None, "traffic_ratio": None, "traffic_source": None, "linear_traffic_GB": None, "linear_traffic_ratio": None}


def roofline(step_fn, steps, cfg_name="c2", step_ms=0.0):
    """HIP-event timing of every library launch over `steps` extra steps (same stream as the launches)."""
    import ctypes
    from mtlora_amd import _lib as L
    lib = L.lib()
    # single stream while profiling: with the per-task head streams on, kernels of different heads overlap and an event
    # bracket then measures the overlapped neighbours too (the timed region above keeps the streams on)
    from mtlora_amd import mtl_harness as H
    streams_on, H._TASK_STREAMS = H._TASK_STREAMS, False
    fstream_on, H._FACTOR_STREAM = H._FACTOR_STREAM, False
    step_fn()
    torch.cuda.synchronize()
    L.check(lib.mtlora_prof_begin(400000), "prof_begin")
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    s = L.ProfSummary()
    L.check(lib.mtlora_prof_end(ctypes.byref(s)), "prof_end")
    H._TASK_STREAMS, H._FACTOR_STREAM = streams_on, fstream_on
    kinds, idx = {}, {}
    for k in range(L.PROF_KINDS):
        if s.count[k]:
            name = lib.mtlora_prof_kind_name(k).decode()
            idx[name] = k
            kinds[name] = {
                "launches_per_step": s.count[k] / steps, "ms_per_step": round(s.ms[k] / steps, 4),
                "avg_us": round(1e3 * s.ms[k] / s.count[k], 2), "launched_GB_per_step": round(s.alg_bytes[k] / steps / 1e9, 4),
                "s8d_GB_per_step": round(s.s8d_bytes[k] / steps / 1e9, 4),
                "GBps_8d": round((s.s8d_bytes[k] / 1e9) / (s.ms[k] / 1e3), 1) if s.ms[k] > 0 else None,
                "TFLOPs": round((s.flops[k] / 1e12) / (s.ms[k] / 1e3), 1) if s.ms[k] > 0 and s.flops[k] > 0 else None}

    def agg(names):
        ks = [idx[n] for n in names if n in idx]
        return (sum(s.count[k] for k in ks), sum(s.ms[k] for k in ks), sum(s.alg_bytes[k] for k in ks),
                sum(s.s8d_bytes[k] for k in ks), sum(s.flops[k] for k in ks))

    nt = ["k_nt:fwd_outputs", "k_nt:fwd_lowrank_P", "k_nt:bwd_lowrank_Q", "k_nt:bwd_dX"]
    n, ms, by, b8, fl = agg(nt)
    ln, lms, lby, lb8, lfl = agg(nt + ["k_tn:dA_dB", "k_pack", "k_tn_reduce", "k_sum"])
    gbs = lambda b, m: (b / 1e9) / (m / 1e3) if m > 0 else 0.0  # noqa: E731
    ach8, achl = gbs(b8, ms), gbs(by, ms)
    tfl = (fl / 1e12) / (ms / 1e3) if ms > 0 else 0.0
    hbm_frac, mfma_frac = ach8 / HBM_PEAK_GBS, tfl / MFMA_PEAK_TFLOPS
    tr = _static_traffic(cfg_name, n / steps, agg(nt + ["k_nt:plain_fwd", "k_nt:plain_dX"])[0] / steps)
    # the same fraction over wider scopes: hot path + factor gradients + pack / reduce + the attention kernels
    # (kernel time), and ALL 8(d) bytes of a step against the step's wall time (everything outside the path counts as time only)
    an, ams, _, ab8, _ = agg(nt + ["k_tn:dA_dB", "k_pack", "k_tn_reduce", "k_sum", "k_attn_fwd", "k_attn_bwd"])
    wider = {"hot_path_with_attention": {"what": "hot-path launches + factor gradients + pack / reduce / sum + window attention fwd / bwd: 8(d) bytes / kernel time",
                                         "launches_per_step": an / steps, "ms_per_step": round(ams / steps, 3), "s8d_GB_per_step": round(ab8 / steps / 1e9, 3),
                                         "GBps": round(gbs(ab8, ams), 1), "frac": round(gbs(ab8, ams) / HBM_PEAK_GBS, 4)}}
    if step_ms:
        wider["whole_step_vs_8d"] = {"what": "all SURVEY 8(d) bytes of one step (MTLoRALinear + attention) / the timed step (glue, heads, losses, optimizer "
                                             "count as time only)", "s8d_GB_per_step": round(ab8 / steps / 1e9, 3), "ms_per_step": round(step_ms, 3),
                                     "GBps": round(gbs(ab8 / steps, step_ms), 1), "frac": round(gbs(ab8 / steps, step_ms) / HBM_PEAK_GBS, 4)}
    return {"bound": "hbm" if hbm_frac >= mfma_frac else "mfma", **wider,
            "kernel": "MTLoRALinear hot-path launches: fused forward / dX (k_sp_xres / k_sp_ares wave-streaming, k_nt / k_ntl / k_ntd tiled) "
                      "and the low-rank P / Q passes that remain (k_sp_proj / k_sp_projsum / k_sp_projk / k_nt)",
            "achieved": round(ach8, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4),
            "frac_achievable": round(ach8 / HBM_ACHIEVABLE_GBS, 4),
            "definition": "SURVEY 8(d) bytes of these launches / their HIP-event time / 8 TB/s (frac_achievable: / 6.29 TB/s)",
            "traffic": tr["traffic"], "traffic_ratio": tr["traffic_ratio"], "traffic_source": tr["traffic_source"],
            "launches_per_step": n / steps, "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
            "alg_bytes_per_launch": round(b8 / max(n, 1)), "kernel_ms_per_step": round(ms / steps, 3),
            "launched": {"achieved": round(achl, 1), "frac": round(achl / HBM_PEAK_GBS, 4),
                         "what": "same launches, fused GELU write / GELU' gate read counted as useful bytes"},
            "mfma": {"achieved": round(tfl, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mfma_frac, 4)},
            "linear_path": {"what": "hot-path launches + factor gradients (k_sp_tn / k_tn) + k_pack + reduce + k_sum vs the whole 8(d) MTLoRALinear bytes",
                            "ms_per_step": round(lms / steps, 3), "s8d_GB_per_step": round(lb8 / steps / 1e9, 3),
                            "GBps": round(gbs(lb8, lms), 1), "frac": round(gbs(lb8, lms) / HBM_PEAK_GBS, 4),
                            "traffic_GB": tr["linear_traffic_GB"], "traffic_ratio": tr["linear_traffic_ratio"],
                            "TFLOPs": round((lfl / 1e12) / (lms / 1e3), 1) if lms > 0 else None},
            "all_kernels": kinds}


def usable_cores():
    """host cores this process may actually use (affinity mask, cgroup quota), not the box's logical CPU count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.999)))
    except Exception:
        pass
    return max(1, min(n, 32))  # beyond ~32 threads the small ATen ops of this model only oversubscribe


def _oracle_step_factory(cfgrow, B, device, amp):
    """the oracle's train step (plain-PyTorch port of the reference) on `device` -- BASELINE legs only."""
    from oracle import mtlora_oracle as O
    tasks = list(cfgrow["tasks"])
    cfg = O.swin_t_cfg(cfgrow["img_size"], tasks, cfgrow["r_shared"], cfgrow["r_task"], embed_dim=cfgrow["embed_dim"],
                       depths=cfgrow["depths"], num_heads=cfgrow["num_heads"], drop_path_rate=0.2)
    shapes = {("backbone." + k): v for k, v in O.backbone_param_shapes(cfg).items()}
    shapes.update(O.head_param_shapes(cfg, {t: O.num_output(t) for t in tasks}))
    P = {k: v.to(device) for k, v in O.make_params(shapes).items()}
    train = [v.requires_grad_(True) for k, v in P.items()
             if O.trainable_filter(k) and not k.endswith(("running_mean", "running_var"))]
    opt = torch.optim.AdamW(train, lr=5e-4, weight_decay=0.05, fused=(device.type == "cuda"))
    img, tg = O.synthetic_batch(B, cfgrow["img_size"], tasks, seed=1234)
    img, tg = img.to(device), {k: v.to(device) for k, v in tg.items()}
    rng = torch.Generator().manual_seed(0)

    def step():
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = O.full_model(P, img, cfg, train=True, rng=rng)
                loss, _ = O.multi_task_loss({k: v.float() for k, v in out.items()}, tg, tasks)
        else:
            loss, _ = O.multi_task_loss(O.full_model(P, img, cfg, train=True, rng=rng), tg, tasks)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in train if p.grad is not None], 5.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
    return step


def eager_leg(cfg_name, B, warm=3, k=10):
    """child process of ``eager_gpu``: time the oracle's ATen dataflow (== the reference's eager PyTorch-ROCm path) and print one JSON
    line.  Runs WITHOUT this file's MIOPEN_FIND_MODE=FAST default (the comparator must not sit on MIOpen's
    fallback solvers): the environment is whatever a user of the reference would have."""
    from mtlora_amd import mtl_harness as H
    row = H.config(cfg_name)
    dev = torch.device("cuda", 0)
    try:
        step = _oracle_step_factory(row, B, dev, amp=True)
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
        res = {"value": round(B / med, 2), "unit": "images/sec", "ms_per_step": round(1e3 * med, 2), "ms_per_step_min": round(1e3 * ts[0], 2),
               "ms_per_step_max": round(1e3 * ts[-1], 2), "value_best": round(B / ts[0], 2), "kind": "port",
               "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE", "default"),
               "sample": f"oracle ATen dataflow on cuda, bf16 autocast, B={B}, own process, {warm} warm-up + {k} timed steps (median; min / max next to it)"}
    except torch.OutOfMemoryError as e:  # the eager path materialises every score / per-task tensor
        res = {"value": None, "error": f"OOM: {str(e)[:80]}"}
    print("EAGER_LEG " + json.dumps(res), flush=True)


def eager_gpu(cfg_name, B, ours_ips):
    """north star comparator: the reference's eager PyTorch-ROCm dataflow on the same GPU / config / batch, in a process of its own
    (clean MIOpen environment, nothing of the product path resident), 3 warm-up + 10 timed steps, median."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "MIOPEN_FIND_MODE"}
    torch.cuda.empty_cache()
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--eager-leg", cfg_name, "--batch", str(B)], env=env,
                             capture_output=True, text=True, timeout=900)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("EAGER_LEG ")]
        if not line:
            return {"value": None, "error": f"eager leg failed (rc {out.returncode}): {out.stderr[-200:]}"}
        res = json.loads(line[-1][len("EAGER_LEG "):])
    except Exception as e:  # noqa: BLE001  (a failing baseline leg must not lose the headline line)
        return {"value": None, "error": f"{type(e).__name__}: {str(e)[:120]}"}
    if res.get("value"):
        res["speedup"] = round(ours_ips / res["value"], 2)
        res["speedup_vs_best_eager_step"] = round(ours_ips / res["value_best"], 2)
    return res


def cpu_baseline(cfgrow):
    """oracle (port of the reference) on the host cores: the config's shapes, B=2, fp32, train mode, bounded sample."""
    n = usable_cores()
    torch.set_num_threads(n)
    B = 2
    step = _oracle_step_factory(cfgrow, B, torch.device("cpu"), amp=False)
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    # ~10-20 s of CPU work in total, bounded whatever the host is
    k = max(2, min(8, int(12.0 / max(warm, 1e-3)))) if warm < 15.0 else (1 if warm < 60.0 else 0)
    if k:
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        dt = (time.perf_counter() - t0) / k
    else:
        dt = warm
    return {"value": round(B / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{cfgrow['what']}: train step, B={B}, fp32, 1 warm-up + {k} timed steps ({dt:.2f} s/step)"}


def run_config(args, name, rank, world, dev, steps, warmup, want_roofline, batch=0):
    """build the config's model, time `steps` train steps, optionally profile the library launches.  Returns (row, B, ips, fields)."""
    from mtlora_amd import mtl_harness as H
    from mtlora_amd.ddp import GradReducer
    row = H.config(name)
    tasks = list(row["tasks"])
    B = batch or row["batch"]
    model = H.build_config_model(name, seed=0, drop_path_rate=0.2).to(dev)
    model.train()
    crit = H.MultiTaskLoss(tasks)
    opt = H.build_optimizer(model, lr=5e-4 * B * world / 512.0,  # main.py:578-583 linear LR scaling
                            capturable=args.graph or args.capturable)
    # replicas start from rank 0's parameters / buffers (GradReducer broadcasts them), not from "every rank seeds 0"
    # ONE bucket for the whole trainable set (33.4 MB at c2; SURVEY 8e: over 7 x 153 GB/s xGMI links the exchange is latency-, not
    # bandwidth-bound, so one large all-reduce lets RCCL pick a direct algorithm instead of several ring passes)
    reducer = (GradReducer(model.parameters(), bucket_mb=34.0, force=args.force_reducer, buffers=model.buffers())
               if (world > 1 or args.force_reducer) else None)
    img, tg = H.synthetic_batch(B, row["img_size"], tasks, seed=1234 + rank, device=dev)
    torch.manual_seed(1234 + rank)

    def eager_step():
        H.train_step(model, crit, opt, img, tg, clip_grad=5.0, reducer=reducer, amp_dtype=torch.bfloat16,
                     fused_loss=not args.no_fused_loss)

    graph_info = {"enabled": False, "why": "eager multi-stream step (default; --graph replays a captured HIP graph)"}
    step = eager_step
    if args.graph:
        # the whole step (fwd + losses + bwd + clip + AdamW) captured as HIP graph(s); RCCL stays outside the graphs
        gstep = H.GraphedTrainStep(model, crit, opt, img, tg, clip_grad=5.0, reducer=reducer, amp_dtype=torch.bfloat16,
                                   fused_loss=not args.no_fused_loss)
        graph_info = {"enabled": gstep.graphed, "why": gstep.why}
        if not gstep.graphed and rank == 0:
            print(f"bench: HIP-graph replay not used, running eagerly: {gstep.why}", file=sys.stderr)
        step = gstep

    dt, t_issue = time_steps(step, steps, warmup, world)
    ips = B * world * steps / dt
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    fields = {
        "value": round(ips, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt / steps, 3),
        "config": {"workload": row["what"] + "; train step (fwd+loss+bwd+clip+AdamW), dropout .05, drop_path .2",
                   "name": name, "per_gpu_batch": B, "global_batch": B * world,
                   "img_size": row["img_size"], "parallelism": f"dp{world}", "trainable_params": n_train,
                   "allreduce_bytes": reducer.nbytes if reducer else 0, "hip_graph": graph_info,
                   "host_issue_ms_per_step": round(1e3 * t_issue / steps, 3), "host_cores": len(os.sched_getaffinity(0))},
    }
    if rank == 0 and want_roofline:
        # profiled EAGERLY: the library's HIP-event brackets are recorded at launch time (same kernels as the graph)
        fields["roofline"] = roofline(eager_step, args.roofline_steps or max(2, min(steps, 5)), name, 1e3 * dt / steps)
    elif want_roofline and world > 1:
        for _ in range(1 + (args.roofline_steps or max(2, min(steps, 5)))):  # keep ranks in lock-step with rank 0's profiled steps
            eager_step()
    del model, opt
    torch.cuda.empty_cache()
    return row, B, ips, fields


def main():
    args = parse()
    if args.eager_leg:
        from mtlora_amd import mtl_harness as H
        return eager_leg(args.eager_leg, args.batch or H.config(args.eager_leg)["batch"])
    full_mask, full_threads = os.sched_getaffinity(0), torch.get_num_threads()
    n_pin = 2 if args.cores < 0 else args.cores
    if n_pin > 0:  # this rank's own host cores (before any thread pool starts)
        allowed = sorted(full_mask)
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        n_pin = max(1, min(n_pin, len(allowed)))
        first = (lr * n_pin) % len(allowed) if len(allowed) >= n_pin * (lr + 1) else 0
        os.sched_setaffinity(0, set(allowed[first:first + n_pin]))
        torch.set_num_threads(n_pin)
    rank, world, local = init_dist(args)
    dev = torch.device("cuda", local)
    from mtlora_amd import _lib as L
    L.lib()  # fail loudly if the HIP extension is missing
    if world > 1 and n_pin <= 0:  # N ranks share one host: keep each rank's intra-op CPU pool small (no CPU-side compute)
        torch.set_num_threads(max(1, min(4, usable_cores() // world)))

    row, B, ips, fields = run_config(args, args.config, rank, world, dev, args.steps, args.warmup, not args.no_roofline, args.batch)
    default = args.config == "c2"
    result = {
        "metric": "images/sec (train step) Swin-T/448 r=64 4-task" if default else f"images/sec (train step) {args.config}",
        "value": fields["value"], "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": fields["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": fields["config"],
    }
    if "roofline" in fields:
        result["roofline"] = fields["roofline"]
    if default and world == 1 and not args.no_other_configs and not args.batch:
        # (before the eager / CPU baseline legs: the CPU leg's thread pools keep spinning and cost the host-heavy c4 step 15 %)
        # the other single-GPU BASELINE configs, short legs (8 steps after 3 warm-up steps, no baselines): Swin-B r=128 and the 8-task r=4 sweep point
        others = {}
        for name in ("c4", "c5:4"):
            try:
                _, Bo, _, f = run_config(args, name, rank, world, dev, 8, 3, not args.no_roofline)
                o = {"value": f["value"], "unit": "images/sec", "ms_per_step": f["ms_per_step"], "per_gpu_batch": Bo,
                     "workload": f["config"]["workload"], "host_issue_ms_per_step": f["config"]["host_issue_ms_per_step"]}
                if "roofline" in f:
                    r = f["roofline"]
                    o["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_achievable", "kernel_ms_per_step",
                                                         "launches_per_step", "alg_bytes_per_launch", "traffic", "traffic_ratio") if k in r}
                    o["roofline"]["mfma_frac"] = r["mfma"]["frac"]
                others[name] = o
            except Exception as e:  # noqa: BLE001  (a failing side leg must not lose the headline line)
                others[name] = {"error": f"{type(e).__name__}: {str(e)[:120]}"}
        result["other_configs"] = others
    if n_pin > 0:  # the comparator legs get the whole host back (the eager leg is a child process and inherits the mask)
        os.sched_setaffinity(0, full_mask)
        torch.set_num_threads(full_threads)
    if rank == 0 and world == 1 and not args.no_eager_gpu:
        result["eager_gpu"] = eager_gpu(args.config, B, ips)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(row)
    if world > 1:
        barrier(world)
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
