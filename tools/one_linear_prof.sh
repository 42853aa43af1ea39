#!/bin/bash
# kernel-level timing of one linear shape (GPU box): tools/one_linear_prof.sh M K N [r] -> per-kernel average ns, default path
# (wave-streaming kernels where eligible) and tiled kernels only (MTLORA_SP=0)
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  rm -rf /tmp/k$m
  MTLORA_SP=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k$m -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/one_linear.py $1 $2 $3 20 ${4:-64} train > /tmp/o$m.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/k$m/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "k_nt" in r["Name"] or "k_sp" in r["Name"]: print("SP=$m  M$1 K$2 N$3 r${4:-64}", r["Name"][:60], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
