#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run (GPU box, through gpurun):  [BENCH_ARGS='--config c4'] tools/kernel_stats.sh <tag>
#   -> gpurun_out/kernel_stats_<tag>.csv   (copy the ones to keep into profiles/)
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
# single stream: per-kernel durations / counters are attributed cleanly only when the heads' task streams do not overlap
export MTLORA_TASK_STREAMS=${MTLORA_TASK_STREAMS:-0}
export MTLORA_FACTOR_STREAM=${MTLORA_FACTOR_STREAM:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eager-gpu --no-roofline --no-other-configs $BENCH_ARGS > /tmp/ks.log 2>&1 || { tail -5 /tmp/ks.log; exit 1; }
mkdir -p $REPO/gpurun_out
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $REPO/gpurun_out/kernel_stats_${TAG}.csv
tail -1 /tmp/ks.log | cut -c1-200
