"""One MTLoRALinear shape, forward only, repeated (profiling target):  python tools/one_linear.py M K N [iters] [r] [train]
`train` (any 6th argument): train mode with lora_dropout 0.05 (the masked projection path) instead of eval."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtlora_amd.lora import MTLoRALinear
M, K, N = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
r = int(sys.argv[5]) if len(sys.argv) > 5 else 64
train = len(sys.argv) > 6
dev = torch.device("cuda")
m = MTLoRALinear(K, N, r={"shared": r}, lora_shared_scale=4.0, lora_dropout=0.05 if train else 0.0, tasks=None).to(dev)
m = m.train() if train else m.eval()
with torch.no_grad():
    m.lora_shared_B.normal_(0, 0.02)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        m(x)
    b.record()
    torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
print(f"M{M} K{K} N{N} r{r}: {ms*1e3:.1f} us/call  {2.0*M*K*N/ms/1e9:.0f} TFLOP/s(base)  {2.0*(M*K+M*N)/ms/1e6:.0f} GB/s(alg)")
