"""GEMM micro-benchmark on the shapes of the GROVE hot path (random data; cdna guide §5.4 rule 25)."""
import sys
import torch
sys.path.insert(0, ".")
from grove_amd import ops

dev = torch.device("cuda:0")
shapes = [
    ("clip fc1", 18464, 4096, 1024), ("clip fc2", 18464, 1024, 4096), ("clip qkv", 18464, 3072, 1024),
    ("llama qkv", 2816, 12288, 4096), ("llama gate/up", 2816, 22016, 4096), ("llama down", 2816, 4096, 11008),
    ("sam qkv win", 56448, 4608, 1280), ("sam lin1", 32768, 5120, 1280), ("sam lin2", 32768, 1280, 5120),
    ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192),
]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for staging in (1, 0):
        ops.gemm_set_staging(bool(staging))
        for _ in range(3):
            ops.linear(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.linear(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res[staging] = 2.0 * M * N * K / ms / 1e9
    print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}  lds-dma {res[1]:7.1f} TF/s   reg-staged {res[0]:7.1f} TF/s", flush=True)
ops.gemm_set_staging(True)
