#!/bin/bash
# GPU busy / idle analysis of the default bench step (all streams on): where does the device wait for the host?
#   tools/timeline.sh [bench args]  ->  gpurun_out/timeline.txt
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-eager-gpu --no-roofline "$@" > /tmp/tl.log 2>&1 || { tail -5 /tmp/tl.log; exit 1; }
F=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
mkdir -p $REPO/gpurun_out
python - "$F" > $REPO/gpurun_out/timeline.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# step boundary = the fused AdamW kernel(s); take the steps between the last 5 optimizer bursts
adam = [i for i, r in enumerate(rows) if "FusedAdam" in r[2]]
bursts = []
for i in adam:
    if not bursts or rows[i][0] - rows[bursts[-1][-1]][1] > 2_000_000: bursts.append([i])
    else: bursts[-1].append(i)
print(f"{len(rows)} dispatches, {len(bursts)} optimizer bursts")
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]
for b0, b1 in list(zip(bursts[:-1], bursts[1:]))[-3:]:
    seg = rows[b0[-1] + 1: b1[-1] + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, n in seg:
        if cur_e is None: cur_s, cur_e, last = s, e, n
        elif s <= cur_e:
            if e > cur_e: cur_e, last = e, n
        else:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last, n, (cur_e - t0) / 1e6))
            cur_s, cur_e, last = s, e, n
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in seg)
    print(f"\nstep: {len(seg)} dispatches, wall {(t1 - t0) / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, sum of kernel durations {ksum / 1e6:.2f} ms")
    hist = collections.Counter()
    for g, a, b, t in gaps:
        hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"] += g
    print("  idle by gap size (ms):", {k: round(v / 1e6, 3) for k, v in hist.items()}, " gaps:", len(gaps))
    # idle by position in the step (10 bins)
    bins = [0.0] * 10
    for g, a, b, t in gaps:
        bins[min(9, int(t / ((t1 - t0) / 1e6) * 10))] += g / 1e6
    print("  idle per tenth of the step (ms):", [round(x, 2) for x in bins])
    print("  largest gaps:")
    for g, a, b, t in sorted(gaps, reverse=True)[:12]:
        print(f"    {g / 1e3:8.1f} us at {t:6.2f} ms   after {short(a)}  ->  {short(b)}")
PY
cat $REPO/gpurun_out/timeline.txt
