"""Fixed-cost / per-K-tile fit of the pipelined 256 x 256 kernel: K sweep, with and without the epilogue (alpha == -12345 skips it)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
M, N = 32768, 5120
for tm in (256, 0):
    L.grove_gemm_set_tile_m(tm)
    for alpha in (1.0, -12345.0):
        if tm == 0 and alpha != 1.0:
            continue
        row = []
        for K in (256, 640, 1280, 2560, 5120):
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            bias = torch.randn(N, device=dev).to(torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            best = 1e9
            for _ in range(3):
                ops.gemm_raw(a, b, out, M, N, K, K, K, N, bias=bias, alpha=alpha)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm_raw(a, b, out, M, N, K, K, K, N, bias=bias, alpha=alpha)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            row.append((K, round(best * 1e3, 1), round(2.0 * M * N * K / best / 1e9)))
        print(f"tile_m={tm} alpha={alpha}: " + "  ".join(f"K={k}: {us} us {tf} TF" for k, us, tf in row), flush=True)
L.grove_gemm_set_tile_m(0)
