#!/usr/bin/env python3
"""P / Q pass time per layer shape (from tools/bench_linear.py --kinds) as one table:  python tools/pq_times.py [--projk N]
 (--projk: functional tuning value for this run: 0 heuristics, 1 tiled only, 2 k_sp_projk whenever, 3 k_pq whenever)"""
import argparse, os, re, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument("--projk", default=None)
a = ap.parse_args()
here = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ)
if a.projk is not None:
    env["MTLORA_SP_PROJK"] = {"0": "1", "1": "0"}.get(a.projk, a.projk)  # (functional._env_tri: "0" -> never, "1"/unset -> heuristics)
SETS = [(["--rs", "128", "--rt", "128"], "b1.qkv b1.fc1 b1.fc2 b2.qkv b2.proj b2.fc1 b2.fc2 b3.qkv b3.fc1 b3.fc2"),
        ([], "s1.fc1 s1.fc2 s2.qkv s2.fc1 s2.fc2 s3.qkv s3.fc1 s3.fc2")]
for extra, shapes in SETS:
    out = subprocess.run([sys.executable, os.path.join(here, "bench_linear.py"), "--kinds", "--knt-only", *extra, "--shapes", *shapes.split()],
                         env=env, capture_output=True, text=True).stdout
    p = q = None
    for line in out.splitlines():
        m = re.search(r"k_nt:fwd_lowrank_P ([0-9.]+)us", line)
        if m: p = float(m.group(1))
        m = re.search(r"k_nt:bwd_lowrank_Q ([0-9.]+)us", line)
        if m: q = float(m.group(1))
        m = re.match(r"^([a-z0-9.A-Z]+)\s+M(\d+) K(\d+) N(\d+)", line)
        if m:
            print(f"{m.group(1):8s} M{m.group(2):>6s} K{m.group(3):>5s} N{m.group(4):>5s}   P {p if p is not None else float('nan'):6.1f} us   Q {q if q is not None else float('nan'):6.1f} us")
            p = q = None
