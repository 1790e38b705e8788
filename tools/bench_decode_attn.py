"""grove_decode_attn alone at the 7B decode shape (B=1, 32 heads x 128, ~650 cached positions): us per launch, eager back to back
and replayed from a HIP graph of 32 launches on 32 different caches (the layers of one token: cold cache rows, as in the step)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
B, H, hd, S, Smax = 1, 32, 128, 650, 720
caches = [torch.randn(B, 2, H, Smax, hd, device=dev).to(bf) for _ in range(32)]
qkv = torch.randn(B, 3 * H * hd, device=dev).to(bf)
pos = torch.full((B,), S, dtype=torch.int32, device=dev)
out = torch.empty(B, H * hd, device=dev, dtype=bf)


def layers():
    for c in caches:
        ops.decode_attn(qkv, c, pos, H, hd, 10000.0, hd ** -0.5, out=out)


layers()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    layers()
for name, fn in (("eager", layers), ("graph", g.replay)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 / 32 * 1e3:.2f} us per launch (32 launches on 32 caches of {S} positions)")
