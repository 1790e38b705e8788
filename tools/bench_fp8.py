"""fp8 GEMM (grove_gemm_fp8) and its activation quantisation on the linear-layer shapes of config 5, next to the bf16 pipelined GEMM."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
shapes = [(5624, 12288, 4096), (5624, 4096, 4096), (5624, 22016, 4096), (5624, 4096, 11008), (36928, 3072, 1024), (36928, 1024, 1024),
          (36928, 4096, 1024), (36928, 1024, 4096), (8192, 8192, 8192)]


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    wq, ws = ops.quant_fp8_rows(w)
    xq = ops.quant_fp8_rows(x)
    out = torch.empty((M, N), dtype=bf, device=dev)
    ms8 = t(lambda: ops.linear_fp8(None, wq, ws, xq=xq, out=out))
    msq = t(lambda: ops.quant_fp8_rows(x))
    ms16 = t(lambda: ops.linear(x, w, out=out))
    fl = 2.0 * M * N * K
    print(f"({M},{N},{K}): fp8 gemm {ms8*1e3:8.1f} us {fl/ms8/1e9:7.1f} TF/s | quant {msq*1e3:6.1f} us | fp8 incl quant {fl/(ms8+msq)/1e9:7.1f} TF/s | bf16 {ms16*1e3:8.1f} us {fl/ms16/1e9:7.1f} TF/s", flush=True)
