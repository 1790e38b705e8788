"""PMC target: the pipelined NT GEMM on the dominant shapes of the step, a few launches each (run under rocprofv3 --pmc)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
shapes = [(32768, 5120, 1280), (32768, 1280, 5120), (56448, 4608, 1280), (2812, 22016, 4096), (2812, 4096, 22016), (56448, 1280, 4608), (2812, 12288, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(a, b, out=out)
    torch.cuda.synchronize()
print("done")
