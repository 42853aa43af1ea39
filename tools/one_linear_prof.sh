#!/bin/bash
# kernel-level timing of one linear shape (GPU box): tools/one_linear_prof.sh M K N [r] -> per-kernel average ns, both MTLORA_NT2 modes
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  rm -rf /tmp/k$m
  MTLORA_NT2=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k$m -- python /root/repo/tools/one_linear.py $1 $2 $3 20 ${4:-64} > /tmp/o$m.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/k$m/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "k_nt" in r["Name"]: print("NT2=$m  M$1 K$2 N$3 r${4:-64}", r["Name"][:48], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
