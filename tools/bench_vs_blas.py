"""Where the hand-written pipelined GEMM stands against the vendor library on the step's plain shapes: torch.mm (hipBLASLt /
rocBLAS under PyTorch-ROCm) vs grove_gemm_bf16, both C = A @ B^T in bf16, interleaved repeats, median. Informational only:
the product path never calls the library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16


def timed(f, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("M N K | grove us (TF/s) | torch.mm us (TF/s)")
for M, N, K in ((32768, 5120, 1280), (32768, 1280, 5120), (32768, 3840, 1280), (18464, 4096, 1024), (18464, 1024, 4096),
                (2812, 22016, 4096), (2812, 4096, 11008), (2812, 12288, 4096), (2812, 4096, 4096), (8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    f1 = lambda: ops.linear(a, w, out=out)
    f2 = lambda: torch.mm(a, w.t(), out=out)
    f1(); f2(); torch.cuda.synchronize()
    r1, r2 = [], []
    for _ in range(5):
        r1.append(timed(f1)); r2.append(timed(f2))
    m1, m2 = sorted(r1)[2], sorted(r2)[2]
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:5d} {K:5d} | {m1:8.1f} ({fl / m1 / 1e6:7.1f}) | {m2:8.1f} ({fl / m2 / 1e6:7.1f})", flush=True)
