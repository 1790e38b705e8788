"""A/B of GEMM tile variants (tile_n, BK) on quantisation-sensitive shapes; interleaved rounds in one process."""
import sys
import torch
sys.path.insert(0, ".")
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
shapes = [(2812, 12288, 4096), (56448, 1280, 4608), (2812, 4096, 22016), (2812, 4096, 4096), (2812, 4096, 11008), (4096, 4096, 4096), (18464, 4096, 1024), (32768, 5120, 1280), (2812, 22016, 4096)]
variants = [("192x128", 128, 192), ("256x128x32", 128, 256), ("auto", 0, 0)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {v[0]: [] for v in variants}
    for rnd_ in range(3):
        for name, tn, bk in variants:
            ops.gemm_set_tile_n(tn)
            _lib.lib().grove_gemm_set_tile_m(bk)
            ops.linear(a, b, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(a, b, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(2.0 * M * N * K / (e0.elapsed_time(e1) / 5) / 1e9)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k}: {max(v):7.1f}" for k, v in res.items()), flush=True)
ops.gemm_set_tile_n(0)
_lib.lib().grove_gemm_set_tile_m(0)
