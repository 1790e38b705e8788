"""GEMV (decode step) microbenchmark: achieved HBM GB/s on the LLaMA-7B projection shapes."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
for M in ([int(a) for a in sys.argv[1:]] or (1, 2)):
    for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32008, 4096)]:
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]  # rotate: defeat the 256 MB MALL
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for w in ws:
            ops.gemv(x, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for w in ws:
                ops.gemv(x, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"M={M} N={N} K={K}: {us:7.1f} us  {N*K*2/us/1e3:7.1f} GB/s", flush=True)
