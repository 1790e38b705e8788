"""PMC target: every stage of the Winograd adapter pipeline at the bench shape (C = 1280, 4 groups x 8 x 32 x 32), 3 launches each."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from grove_amd import ops
dev = torch.device("cuda:0")
C, geom = 1280, (4, 8, 32, 32)
rows, tiles, bf = 32768, 4096, torch.bfloat16
x = torch.randn(rows, C, device=dev).to(bf); dz = torch.randn(rows, C, device=dev).to(bf)
w = (torch.randn(C, 27 * C, device=dev) * 0.02).to(bf); b = torch.randn(C, device=dev).to(bf); a = torch.tensor([0.1], device=dev)
V = torch.empty(64, tiles, C, dtype=bf, device=dev); dM = torch.empty_like(V); Mh = torch.empty_like(V)
U = torch.empty(64, C, C, dtype=bf, device=dev); dU = torch.empty(64, C, C, dtype=torch.float32, device=dev)
y, pre = torch.empty_like(x), torch.empty_like(x); gw = torch.zeros(C, 27 * C, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.wino3d_transform_tokens(x, geom, 0, out=V)
for _ in range(3):
    ops.wino3d_transform_tokens(dz, geom, 1, out=dM)
for _ in range(3):
    ops.wino3d_transform_weight(w, out=U)
for _ in range(3):
    ops.gemm_raw(V, U, Mh, 64 * tiles, C, C, C, C, C, b_group=tiles)
for _ in range(3):
    ops.wino3d_output(Mh, geom, y, bias=b, act=ops.ACT_RELU, scale_ptr=a, scale_tanh=True, residual=x, aux=pre)
for _ in range(3):
    ops.wgrad(dM, V, dU, K=tiles, k_batches=64, sC_batch=C * C, overwrite=True, M=C, N=C)
for _ in range(3):
    ops.wino3d_wgrad_output(dU, gw, scale_ptr=a, scale_tanh=True)
torch.cuda.synchronize()
print("done")
