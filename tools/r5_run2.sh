set -x
mkdir -p gpurun_out/r5
rm -f gpurun_out/kernel_parity.jsonl gpurun_out/model_parity.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=40 2>&1 | tail -70 > gpurun_out/r5/pytest_gpu_full.txt
tail -60 gpurun_out/r5/pytest_gpu_full.txt
