#!/bin/bash
# End-of-round measurement set (GPU box, through gpurun; ~15 min): bash tools/final_profiles.sh [part]   -> gpurun_out/final/
# part 1: tests + bench lines;  part 2: rocprofv3 kernel stats + PMC passes + per-layer tables
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
PART=${1:-1}
if [ "$PART" = "1" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1; grep -E "passed|failed" $OUT/pytest.txt | tail -1
  cp gpurun_out/model_parity.jsonl $OUT/model_parity.jsonl 2>/dev/null
  python bench.py > $OUT/bench_default.json 2>/dev/null; cut -c1-160 $OUT/bench_default.json
  for c in c1 c4 c5:4 c5:16 c5:64 c5:256; do
    python bench.py --config $c --no-cpu-baseline --no-other-configs > $OUT/bench_${c/:/_}.json 2>/dev/null; cut -c1-120 $OUT/bench_${c/:/_}.json
  done
  python bench.py --config c1 --graph --no-cpu-baseline --no-eager-gpu --no-other-configs > $OUT/bench_c1_graph.json 2>/dev/null; cut -c1-120 $OUT/bench_c1_graph.json
  python bench.py --cores 2 --no-cpu-baseline --no-eager-gpu --no-other-configs --no-roofline > $OUT/bench_c2_cores2_eager.json 2>/dev/null; cut -c1-120 $OUT/bench_c2_cores2_eager.json
  python bench.py --cores 2 --graph --no-cpu-baseline --no-eager-gpu --no-other-configs --no-roofline > $OUT/bench_c2_cores2_graph.json 2>/dev/null; cut -c1-120 $OUT/bench_c2_cores2_graph.json
else
  bash tools/kernel_stats.sh c2 && cp gpurun_out/kernel_stats_c2.csv $OUT/
  BENCH_ARGS="--config c4" bash tools/kernel_stats.sh c4 && cp gpurun_out/kernel_stats_c4.csv $OUT/
  # HBM traffic per launch KIND (KINDS=1: the last step runs under the library's launch profiler; tools/pmc_traffic.py zips its record
  # list with the per-dispatch counters: hot-path launches vs the callers' rank-0 GEMMs on the same kernels)
  KINDS=1 bash tools/pmc.sh FETCH_SIZE fetch && cp gpurun_out/pmc_fetch*.csv $OUT/
  KINDS=1 bash tools/pmc.sh WRITE_SIZE write && cp gpurun_out/pmc_write*.csv $OUT/
  python tools/pmc_traffic.py gpurun_out/pmc_fetch.csv gpurun_out/pmc_write.csv 4 r06_pmc_fetch.csv r06_pmc_write.csv > $OUT/pmc_traffic.json
  KINDS=1 BENCH_ARGS="--config c4" bash tools/pmc.sh FETCH_SIZE fetch_c4 && cp gpurun_out/pmc_fetch_c4*.csv $OUT/
  KINDS=1 BENCH_ARGS="--config c4" bash tools/pmc.sh WRITE_SIZE write_c4 && cp gpurun_out/pmc_write_c4*.csv $OUT/
  python tools/pmc_traffic.py gpurun_out/pmc_fetch_c4.csv gpurun_out/pmc_write_c4.csv 4 r06_pmc_fetch_c4.csv r06_pmc_write_c4.csv > $OUT/pmc_traffic_c4.json
  # the static traffic figure bench.py quotes is only honest while the profiled launch structure IS the benched one
  python - $OUT/pmc_traffic.json <<'PY' || { echo "pmc_traffic.json: hot-path launch count differs from bench.py's -- re-run tools/pmc.sh after the kernel change"; exit 1; }
import json, subprocess, sys
pm = json.load(open(sys.argv[1]))
line = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-eager-gpu", "--no-other-configs"],
                      capture_output=True, text=True).stdout.strip().splitlines()[-1]
n_bench = json.loads(line)["roofline"]["launches_per_step"]
n_pmc = pm["hot_path"]["launches_per_step"]
print("hot-path launches per step: bench", n_bench, "pmc", n_pmc)
sys.exit(0 if abs(n_bench - n_pmc) < 0.5 else 1)
PY
  bash tools/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" sq && cp gpurun_out/pmc_sq.csv $OUT/
  cd $REPO
  python tools/bench_linear.py --kinds > $OUT/bench_linear.txt 2>&1
  bash tools/tn_ab.sh > $OUT/tn_ab.txt 2>&1
  bash tools/ntd_ab.sh > $OUT/ntd_ab.txt 2>&1
  bash tools/projk_ab.sh > $OUT/projk_ab.txt 2>&1
  python tools/pq_times.py > $OUT/pq_times.txt 2>&1
  python tools/phase_times.py > $OUT/phase_times.txt 2>&1
  python tools/host_breakdown.py > $OUT/host_breakdown.txt 2>&1
  tools/probe/wprobe > $OUT/wprobe.txt 2>&1
  python tools/bw_probe.py > $OUT/bw_probe.txt 2>&1
  ls -la $OUT
fi
