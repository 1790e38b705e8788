"""PMC target: the persistent NT GEMM instances of the step on one shape each (3 launches): instruction mix and wait shares per instance."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from grove_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
def run(M, N, K, act=ops.ACT_NONE, aux=False, res=False):
    x = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.03).to(bf); b = torch.randn(N, device=dev).to(bf)
    r = torch.randn(M, N, device=dev).to(bf) if res else None
    a = torch.empty(M, N, device=dev, dtype=bf) if aux else None
    for _ in range(3):
        ops.linear(x, w, b if act != ops.ACT_NONE else None, act=act, residual=r, aux=a)
run(32768, 1280, 5120, res=True)            # <256, false, -1>: SAM fc2
run(32768, 5120, 1280, ops.ACT_GELU, aux=True)  # <256, false, GELU>: SAM fc1
run(2812, 4096, 11008, res=True)            # <192, false, -1>: LLaMA down (schedule 2)
run(18464, 4096, 1024, ops.ACT_QUICKGELU)   # <256, false, QuickGELU>: CLIP fc1
torch.cuda.synchronize()
print("done")
