"""A/B of the 256x256 pipelined NT GEMM against the auto-selected tiles; checks results first."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
shapes = [(2812, 12288, 4096), (56448, 1280, 4608), (2812, 4096, 22016), (2812, 4096, 4096), (2812, 4096, 11008), (4096, 4096, 4096),
          (18464, 4096, 1024), (32768, 5120, 1280), (32768, 1280, 5120), (32768, 3840, 1280), (2812, 22016, 4096), (8192, 8192, 8192), (1000, 520, 192), (300, 260, 64), (777, 1000, 128), (4096, 4096, 320)]
variants = [("auto", 0), ("pp256", 256), ("c256", 257)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    outs = {}
    res = {v[0]: [] for v in variants}
    for rnd_ in range(3):
        for name, tm in variants:
            L.grove_gemm_set_tile_m(tm)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.linear(a, b, bias, out=out)
            torch.cuda.synchronize()
            outs[name] = out
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(a, b, bias, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(2.0 * M * N * K / (e0.elapsed_time(e1) / 5) / 1e9)
    err = max((outs["auto"].float() - outs[k].float()).abs().max().item() for k in ("pp256", "c256"))
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k}: {max(v):7.1f}" for k, v in res.items()) + f"  maxdiff {err:.3g}", flush=True)
L.grove_gemm_set_tile_m(0)
