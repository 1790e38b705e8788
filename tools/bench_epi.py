"""Is the pipelined GEMM's epilogue bound per CU or by the chip-wide write burst? Time forced 256x256 launches with few / many tiles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
L.grove_gemm_set_tile_m(256)
def run(M, N, K, reps=20):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    f = lambda: ops.linear(a, w)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = (tiles + 255) // 256
    print(f"M={M:6d} N={N:5d} K={K:5d} tiles={tiles:5d} rounds={rounds:3d}: {us:8.1f} us  {us / rounds:7.2f} us/round  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
def timed(f, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
STS = (0,)
print("median us per launch by stagger", STS)
for M, N, K in ((32768, 5120, 1280), (32768, 1280, 5120), (32768, 3840, 1280), (32768, 1280, 1280), (18464, 4096, 1024), (18464, 1024, 4096),
                (18464, 3072, 1024), (18464, 1024, 1024), (2812, 22016, 4096), (2812, 4096, 22016), (2812, 12288, 4096), (2812, 4096, 4096), (32768, 1024, 128)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    f = lambda: ops.linear(a, w)
    for tm in (256, 193):
        L.grove_gemm_set_tile_m(tm)
        f(); torch.cuda.synchronize()
        res = {st: [] for st in STS}
        for rep in range(5):
            for st in STS:
                L.grove_gemm_set_stagger(st)
                res[st].append(timed(f))
        med = [sorted(res[st])[2] for st in STS]
        flops = 2.0 * M * N * K
        bm = 256 if tm == 256 else 192
        tiles = ((M + bm - 1) // bm) * ((N + 255) // 256)
        print(f"M={M:6d} N={N:5d} K={K:5d} bm={bm} rounds={(tiles + 255) // 256:2d}: " + " ".join(f"{m:7.1f}" for m in med) + f"  us   {flops / med[0] / 1e6:7.1f} TF/s", flush=True)
