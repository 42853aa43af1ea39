#!/bin/bash
# PMC pass over a short bench run (run on the GPU box through gpurun; counters in their own run, kernel-trace only):
#   tools/pmc.sh "<counters>" <tag>  ->  gpurun_out/pmc_<tag>.csv  (per kernel name: dispatches, sum of each counter)
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CTRS="$1"; TAG=${2:-pmc}
# single stream: per-kernel durations / counters are attributed cleanly only when the heads' task streams do not overlap
export MTLORA_TASK_STREAMS=${MTLORA_TASK_STREAMS:-0}
export MTLORA_FACTOR_STREAM=${MTLORA_FACTOR_STREAM:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pm -- ${PMC_CMD:-python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eager-gpu --no-roofline --no-other-configs $BENCH_ARGS} > /tmp/pm.log 2>&1 || { tail -5 /tmp/pm.log; exit 1; }
F=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
mkdir -p $REPO/gpurun_out
python - "$F" "$REPO/gpurun_out/pmc_${TAG}.csv" <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"][:70].replace(",", ";")
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); disp[n].add(r["Dispatch_Id"])
ctrs = sorted({c for v in agg.values() for c in v})
with open(sys.argv[2], "w") as f:
    f.write("name,dispatches," + ",".join(ctrs) + "\n")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        f.write(n + "," + str(len(disp[n])) + "," + ",".join("%.0f" % v.get(c, 0) for c in ctrs) + "\n")
PY
wc -l $REPO/gpurun_out/pmc_${TAG}.csv
