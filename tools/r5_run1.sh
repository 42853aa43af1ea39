set -x
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "block_call" 2>&1 | tail -15 > gpurun_out/r5/t_block.txt
cat gpurun_out/r5/t_block.txt
for f in 0 1; do
  MTLORA_FUSED_BLOCKS=$f python tools/host_breakdown.py --config c2 > gpurun_out/r5/host_c2_f$f.txt 2>&1
  MTLORA_FUSED_BLOCKS=$f python tools/host_breakdown.py --config c4 > gpurun_out/r5/host_c4_f$f.txt 2>&1
done
head -12 gpurun_out/r5/host_c2_f0.txt gpurun_out/r5/host_c2_f1.txt gpurun_out/r5/host_c4_f0.txt gpurun_out/r5/host_c4_f1.txt
for f in 0 1; do
  MTLORA_FUSED_BLOCKS=$f python bench.py --no-other-configs --no-eager-gpu --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5/b_c2_f$f.json 2>gpurun_out/r5/b_c2_f$f.err
  MTLORA_FUSED_BLOCKS=$f python bench.py --config c4 --no-eager-gpu --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5/b_c4_f$f.json 2>gpurun_out/r5/b_c4_f$f.err
done
python - <<'PY'
import json
for n in ("c2_f0","c2_f1","c4_f0","c4_f1"):
    try:
        d=json.loads(open(f"gpurun_out/r5/b_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["host_issue_ms_per_step"], d.get("roofline",{}).get("kernel_ms_per_step"))
    except Exception as e:
        print(n, "ERR", e)
PY
