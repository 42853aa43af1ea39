timeout 900 python -m pytest tests -q -m gpu -k "attention or pending or window_attention or swin_block" 2>&1 | grep -E "passed|failed|^E " | tail -12
python tests/perf_kernels.py --only attn 2>&1 | tail -12
