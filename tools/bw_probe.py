"""HBM ceilings with ATen's own streaming kernels (context for the roofline fractions): pure write (fill), pure read (sum),
copy (read + write), 1:3 read:write mix shaped like a K = 96 -> N = 288 layer."""
import torch
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 401408
y = torch.empty(M, 288, device=dev, dtype=torch.bfloat16)
x = torch.randn(M, 96, device=dev, dtype=torch.bfloat16)
z = torch.empty_like(y)
ms = t(lambda: y.fill_(1.0)); print(f"fill  {y.numel()*2/1e6:.0f} MB: {ms*1e3:.1f} us  {y.numel()*2/ms/1e9:.2f} TB/s")
ms = t(lambda: y.sum()); print(f"sum   {y.numel()*2/1e6:.0f} MB: {ms*1e3:.1f} us  {y.numel()*2/ms/1e9:.2f} TB/s")
ms = t(lambda: z.copy_(y)); print(f"copy  {2*y.numel()*2/1e6:.0f} MB: {ms*1e3:.1f} us  {2*y.numel()*2/ms/1e9:.2f} TB/s")
y3 = y.view(M, 3, 96)
ms = t(lambda: torch.mul(x.unsqueeze(1), 2.0, out=None).expand(M, 3, 96).contiguous()); print(f"1r:3w (2 kernels) {ms*1e3:.1f} us")
def mix():
    y3.copy_(x.unsqueeze(1).expand(M, 3, 96))
ms = t(mix); print(f"1r:3w copy-expand {(x.numel()+y.numel())*2/1e6:.0f} MB: {ms*1e3:.1f} us  {(x.numel()+y.numel())*2/ms/1e9:.2f} TB/s")
