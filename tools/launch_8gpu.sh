#!/bin/bash
# The multi-GPU scaling bench as ONE command per N (one process per GPU, RCCL over xGMI, weak scaling: 32 images per GPU):
#     bash tools/launch_8gpu.sh [N=8] [steps=30] [warmup=10] [extra bench.py flags, e.g. --cores 2]
# N = 1 runs bench.py directly; N > 1 is exactly the line the driver uses (SURVEY 8e / BASELINE configs[2]).  Each rank prints nothing but
# rank 0's ONE JSON line; `value` is the whole-job images/sec, `config.allreduce_bytes` the 33.4 MB of trainable gradients per step.
# Host budget: the step is GPU-bound down to 2 host cores per rank; on a node with fewer than 4 cores per rank pass
# `--cores 1` (each rank pins itself to its own cores [r N, (r + 1) N) of the affinity mask; default 2 per rank).
N=${1:-8}; STEPS=${2:-30}; WARMUP=${3:-10}; shift 3 2>/dev/null
REPO=$(cd "$(dirname "$0")/.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC only on this driver: RCCL needs it for the peer mappings
cd "$REPO"
if [ "$N" = "1" ]; then
  exec python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" --no-other-configs "$@"
fi
PORT=${MASTER_PORT:-$((29500 + RANDOM % 400))}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
     bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" "$@"
