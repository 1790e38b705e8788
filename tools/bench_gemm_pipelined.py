"""Epilogue variants of the pipelined kernel vs the 128-row kernels: plain / bias+GELU+aux / bias+quickGELU+aux / residual."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
def run(M, N, K, tm, **kw):
    L.grove_gemm_set_tile_m(tm)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    args = dict(bias=bias)
    if kw.get("aux"):
        args["aux"] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if kw.get("res"):
        args["residual"] = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if kw.get("act"):
        args["act"] = kw["act"]
    best = 1e9
    for _ in range(3):
        ops.linear(a, b, out=out, **args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.linear(a, b, out=out, **args)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best, out, args.get("aux")
for (M, N, K) in [(32768, 5120, 1280), (18464, 4096, 1024), (32768, 1280, 5120)]:
    for name, kw in [("plain", {}), ("gelu+aux", dict(act=ops.ACT_GELU, aux=True)), ("qgelu+aux", dict(act=ops.ACT_QUICKGELU, aux=True)), ("residual", dict(res=True))]:
        torch.manual_seed(0)
        t0, o0, x0 = run(M, N, K, 128, **kw)
        torch.manual_seed(0)
        t1, o1, x1 = run(M, N, K, 0, **kw)
        d = (o0.float() - o1.float()).abs().max().item()
        dx = (x0.float() - x1.float()).abs().max().item() if x0 is not None else 0.0
        print(f"M={M} N={N} K={K} {name:10s}: 128-row {t0:7.1f}us  auto {t1:7.1f}us  maxdiff {d:.3g} aux {dx:.3g}", flush=True)
L.grove_gemm_set_tile_m(0)
