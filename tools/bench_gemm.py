"""hipBLASLt (torch.addmm, bf16) on the GEMM shapes of the MTLoRALinear layers -- how far is k_nt from a tuned dense GEMM at
each stage?   python tools/bench_gemm.py"""
import torch, time
dev = torch.device("cuda", 0)
shapes = [  # (M, K, N) forward ; the dX GEMM is (M, N, K)
    (401408, 96, 288), (401408, 96, 384), (401408, 384, 96), (401408, 160, 288),
    (100352, 192, 576), (100352, 192, 768), (100352, 768, 192), (100352, 256, 768),
    (25088, 384, 1152), (25088, 384, 1536), (25088, 1536, 384), (25088, 448, 1536), (25088, 1600, 384),
    (6272, 768, 2304), (6272, 768, 3072), (6272, 3072, 768), (6272, 832, 3072),
]
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        y = torch.addmm(b, x, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = torch.addmm(b, x, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * M * K * N
    by = 2.0 * (M * K + M * N + N * K)
    print(f"M{M:7d} K{K:5d} N{N:5d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e6:5.2f} TB/s")
