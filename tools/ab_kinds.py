#!/usr/bin/env python3
"""Per-kind kernel time of two bench.py lines side by side:  python tools/ab_kinds.py a.json b.json"""
import json, sys
a, b = (json.loads(open(p).read().strip().splitlines()[-1]) for p in sys.argv[1:3])
print(f"{'':24s} {'A':>9s} {'B':>9s}   (ms/step)   step: {a['ms_per_step']:.2f} vs {b['ms_per_step']:.2f}   {a['value']:.1f} vs {b['value']:.1f}")
ka, kb = a["roofline"]["all_kernels"], b["roofline"]["all_kernels"]
ta = tb = 0.0
for k in ka:
    x, y = ka[k]["ms_per_step"], kb.get(k, {}).get("ms_per_step", float("nan"))
    ta += x; tb += y
    print(f"{k:24s} {x:9.3f} {y:9.3f}  {y - x:+.3f}")
print(f"{'sum':24s} {ta:9.3f} {tb:9.3f}  {tb - ta:+.3f}")
