#!/bin/bash
# PMC counters of an arbitrary command (GPU box): tools/pmc_cmd.sh "<counters>" <tag> <cmd...>  -> gpurun_out/pmc_<tag>.csv
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CTRS="$1"; TAG="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$TAG
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pm_$TAG -- "$@" > /tmp/pm_$TAG.log 2>&1 || { tail -5 /tmp/pm_$TAG.log; exit 1; }
F=$(find /tmp/pm_$TAG -name '*counter_collection.csv' | head -1)
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
head -4 $REPO/gpurun_out/pmc_${TAG}.csv
