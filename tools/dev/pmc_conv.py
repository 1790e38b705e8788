"""PMC target: the Conv3d adapter's three GEMMs at the step's size — NT forward (gathered A, tap skip), NT dgrad, TN weight gradient
(gathered B, tap skip) — 3 launches each (fabric fetch / write counters: why does the TN kernel lose twice as much to its LDS-DMA?)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from grove_amd import ops
from grove_amd.model.indexing import conv3d_gather_index
dev = torch.device("cuda:0")
bf = torch.bfloat16
G, T, H, W, C = 4, 8, 32, 32, 1280
M = G * T * H * W
x = torch.randn(M, C, device=dev).to(bf)
w = (torch.randn(C, 27 * C, device=dev) * 0.02).to(bf)
dy = torch.randn(M, C, device=dev).to(bf)
idx = conv3d_gather_index(G, T, H, W).to(dev)
for _ in range(3):
    ops.linear(x, w, a_idx=idx, a_taps=27, a_frames=(H * W, T))
o = torch.zeros(C, 27 * C, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.wgrad(dy, x, o, b_idx=idx, b_taps=27, b_frames=(H * W, T))
x27 = torch.randn(M, 27 * C, device=dev).to(bf)
for _ in range(3):
    ops.wgrad(dy, x27, o)
torch.cuda.synchronize()
print("done")
