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
mkdir -p $REPO/gpurun_out
# KINDS=1: the LAST step of the run is executed under the library's own launch profiler (bench.py's roofline leg, one step) and its
# record list (one line per library launch in issue order: kind, bytes, tag) is kept next to a PER-DISPATCH counter table, so that
# tools/pmc_traffic.py can attribute the counters to launch kinds (hot-path launches vs the callers' rank-0 GEMMs of the same kernels)
if [ "${KINDS:-0}" = "1" ]; then
  export MTLORA_PROF_DUMP=$REPO/gpurun_out/pmc_${TAG}_kinds.csv
  ROOF="--roofline-steps 1"
else
  ROOF="--no-roofline"
fi
timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pm -- ${PMC_CMD:-python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eager-gpu $ROOF --no-other-configs $BENCH_ARGS} > /tmp/pm.log 2>&1 || { tail -5 /tmp/pm.log; exit 1; }
F=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
python - "$F" "$REPO/gpurun_out/pmc_${TAG}.csv" "$REPO/gpurun_out/pmc_${TAG}_dispatch.csv" <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set); per = defaultdict(lambda: defaultdict(float)); nm = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"][:70].replace(",", ";")
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); disp[n].add(r["Dispatch_Id"])
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); nm[int(r["Dispatch_Id"])] = n
ctrs = sorted({c for v in agg.values() for c in v})
with open(sys.argv[2], "w") as f:
    f.write("name,dispatches," + ",".join(ctrs) + "\n")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        f.write(n + "," + str(len(disp[n])) + "," + ",".join("%.0f" % v.get(c, 0) for c in ctrs) + "\n")
with open(sys.argv[3], "w") as f:  # one line per dispatch, in dispatch order
    f.write("dispatch,name," + ",".join(ctrs) + "\n")
    for d in sorted(per):
        f.write(str(d) + "," + nm[d] + "," + ",".join("%.0f" % per[d].get(c, 0) for c in ctrs) + "\n")
PY
wc -l $REPO/gpurun_out/pmc_${TAG}.csv
