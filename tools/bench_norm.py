"""Norm forward / backward streams at the step's shapes: time and achieved HBM bandwidth (algorithmic bytes)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def timeit(f, n=20):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for name, rows, C, rms in (("SAM LayerNorm", 32768, 1280, False), ("CLIP LayerNorm", 18464, 1024, False), ("LLaMA RMSNorm", 2812, 4096, True)):
    x = torch.randn(rows, C, device=dev).to(bf16)
    dy = torch.randn(rows, C, device=dev).to(bf16)
    w = torch.randn(C, device=dev).to(bf16)
    b = torch.randn(C, device=dev).to(bf16)
    dx = torch.empty_like(x)
    perm = torch.randperm(rows, device=dev).to(torch.int32)
    if rms:
        t = timeit(lambda: ops.rmsnorm_bwd(x, w, dy, 1e-5, dx=dx))
        print(f"{name} [{rows}, {C}] bwd: {t:7.1f} us  {3 * rows * C * 2 / t / 1e6:.2f} TB/s")
    else:
        mean = x.float().mean(1)
        rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt()
        t = timeit(lambda: ops.layernorm_bwd(x, w, dy, mean, rstd, dx=dx))
        print(f"{name} [{rows}, {C}] bwd: {t:7.1f} us  {3 * rows * C * 2 / t / 1e6:.2f} TB/s")
        t = timeit(lambda: ops.layernorm_bwd(x, w, dy, mean, rstd, dx=dx, in_idx=perm))
        print(f"{name} [{rows}, {C}] bwd, gathered dy rows: {t:7.1f} us  {3 * rows * C * 2 / t / 1e6:.2f} TB/s")
        t = timeit(lambda: ops.layernorm_bwd(x, w, dy, mean, rstd, dx=dx, accumulate=True))
        print(f"{name} [{rows}, {C}] bwd, dx +=: {t:7.1f} us  {4 * rows * C * 2 / t / 1e6:.2f} TB/s")
