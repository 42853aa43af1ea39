#!/bin/bash
# per-launch time of the fused wave-streaming forward with parts of it switched off (MTLORA_SP_DBG bits: 1 no output stores,
# 2 no slab loads, 4 no block MFMAs, 8 no P store).  Usage: tools/sp_ablate.sh s0.qkv [s0.fc1 ...]
for sh in "$@"; do
  for d in 0 1 2 4 8 9 11 15; do
    echo -n "$sh dbg=$d: "
    MTLORA_SP_DBG=$d python tools/bench_linear.py --knt-only --kinds --shapes $sh 2>/dev/null | grep fwd_outputs | head -1
  done
done
