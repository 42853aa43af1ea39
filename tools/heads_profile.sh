#!/bin/bash
# per-kernel time of the heads alone (one stream):  tools/heads_profile.sh [tag]  ->  gpurun_out/heads_stats_<tag>.csv
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -- python $REPO/tools/heads_profile.py --iters 5 > /tmp/hp.log 2>&1 || { tail -5 /tmp/hp.log; exit 1; }
mkdir -p $REPO/gpurun_out
F=$(find /tmp/hp -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp "$F" $REPO/gpurun_out/heads_stats_${TAG}.csv
grep "heads fwd" /tmp/hp.log
