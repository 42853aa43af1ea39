#!/bin/bash
# the driver's GPU tier, as the driver runs it (through gpurun):  bash tools/gpu_suite.sh  ->  gpurun_out/pytest_gpu.txt (wall time, 40 slowest)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
rm -f gpurun_out/kernel_parity.jsonl gpurun_out/model_parity.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 2>&1 | tail -45 > gpurun_out/pytest_gpu.txt
grep -E "passed|failed" gpurun_out/pytest_gpu.txt | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
