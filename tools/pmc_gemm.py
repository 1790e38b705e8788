"""PMC target: the pipelined NT GEMM on the dominant shapes of the step, a few launches each (run under rocprofv3 --pmc)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
# the plain-epilogue launches that carry most of the step's GEMM time (profiles/r01_v9_gemm_shapes.txt): 256-row instance, then 192-row
shapes = [(32768, 1280, 5120), (32768, 5120, 1280), (32768, 3840, 1280), (2812, 11008, 4096), (32768, 1280, 1280),
          (2812, 4096, 22016), (2812, 12288, 4096), (2812, 4096, 11008), (2812, 4096, 4096), (18464, 1024, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(a, b, out=out)
    torch.cuda.synchronize()
print("done")
