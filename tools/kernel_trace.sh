#!/bin/bash
# Per-dispatch kernel trace of one bench step (run on the GPU box through gpurun):
#   tools/kernel_trace.sh [name-filter]  ->  gpurun_out/trace_<filter>.csv  (name, dur_us, grid, wg, lds, vgpr)
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
FILT=${1:-k_}
# single stream: per-kernel durations / counters are attributed cleanly only when the heads' task streams do not overlap
export MTLORA_TASK_STREAMS=${MTLORA_TASK_STREAMS:-0}
export MTLORA_FACTOR_STREAM=${MTLORA_FACTOR_STREAM:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-gpu --no-roofline > /tmp/tr.log 2>&1 || { tail -5 /tmp/tr.log; exit 1; }
F=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
mkdir -p $REPO/gpurun_out
python - "$F" "$FILT" "$REPO/gpurun_out/trace_${FILT}.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last third of the dispatches (= the last timed step)
rows = rows[len(rows) * 2 // 3:]
with open(sys.argv[3], "w") as f:
    f.write("name,dur_us,grid_x,grid_y,grid_z,wg,lds,vgpr,scratch\n")
    for r in rows:
        if sys.argv[2] in r["Kernel_Name"]:
            f.write("%s,%.2f,%s,%s,%s,%s,%s,%s,%s\n" % (r["Kernel_Name"][:140].replace(",", ";"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                    r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Scratch_Size"]))
PY
wc -l $REPO/gpurun_out/trace_${FILT}.csv
