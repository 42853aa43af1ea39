#!/bin/bash
# needs a library built with the toggles compiled in:  MTLORA_ABLATE=1 python -m mtlora_amd.csrc.build --force
# ablation timing of the lean NT kernels (results are WRONG with any bit set): MTLORA_NT_DBG bits
#   1 no global stores, 2 no global loads, 4 no MFMA, 8 no epilogue, 16 return at once, 32 no affine (k_nt), 64 no LDS staging
#   stores, 128 no barriers (k_nt)
for d in ${DBGS:-0 15 79}; do echo "== MTLORA_NT_DBG=$d (tiled kernels: MTLORA_SP=0)"; MTLORA_SP=0 MTLORA_NT_DBG=$d python tools/bench_linear.py --shapes ${SHAPES:-s0.qkv s0.fc2 s1.fc1} --kinds --knt-only 2>&1 | grep -v amdgpu.ids; done
