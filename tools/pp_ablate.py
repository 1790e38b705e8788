"""Times the 256-row pipelined GEMM of the library in GROVE_HIP_LIB on a few large shapes (ablation builds: results may be wrong)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.lib()
L.grove_gemm_set_tile_m(256)
out_line = []
for M, N, K in [(32768, 1280, 5120), (32768, 5120, 1280), (2816, 4096, 22016), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    best = 1e9
    for _ in range(3):
        ops.linear(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    out_line.append(f"({M},{N},{K}) {best:7.1f}us {2.0 * M * N * K / best / 1e6:5.0f}TF")
print(" | ".join(out_line))
