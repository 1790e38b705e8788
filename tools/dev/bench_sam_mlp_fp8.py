"""SAM block MLP (lin1 + GELU, lin2) at config 5's row count: bf16 GEMMs vs e4m3 GEMMs + their activation quantisation passes."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, n=3):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


C = 1280
for M in (32768, 65536):
    h = torch.randn(M, C, device=dev).to(bf)
    w1 = (torch.randn(4 * C, C, device=dev) * 0.02).to(bf); b1 = torch.randn(4 * C, device=dev).to(bf)
    w2 = (torch.randn(C, 4 * C, device=dev) * 0.02).to(bf); b2 = torch.randn(C, device=dev).to(bf)
    w1q, w2q = ops.quant_fp8_rows(w1), ops.quant_fp8_rows(w2)
    f = ops.linear(h, w1, b1, act=ops.ACT_GELU)
    hq, fq = ops.quant_fp8_rows(h), ops.quant_fp8_rows(f)
    fl = 2.0 * M * C * 4 * C
    r = {"bf16 lin1+gelu": timeit(lambda: ops.linear(h, w1, b1, act=ops.ACT_GELU)), "bf16 lin2": timeit(lambda: ops.linear(f, w2, b2)),
         "quant h": timeit(lambda: ops.quant_fp8_rows(h)), "quant f": timeit(lambda: ops.quant_fp8_rows(f)),
         "fp8 lin1+gelu (codes given)": timeit(lambda: ops.linear_fp8(h, w1q[0], w1q[1], b1, act=ops.ACT_GELU, xq=hq)),
         "fp8 lin1 no act (codes given)": timeit(lambda: ops.linear_fp8(h, w1q[0], w1q[1], b1, xq=hq)),
         "quant gelu(pre)": timeit(lambda: ops.quant_fp8_rows(f, act=ops.ACT_GELU)),
         "fp8 lin2 (codes given)": timeit(lambda: ops.linear_fp8(f, w2q[0], w2q[1], b2, xq=fq))}
    print(f"M = {M}")
    for k, v in r.items():
        print(f"  {k:32s} {v:8.1f} us" + (f"  {fl / v / 1e6:7.1f} TF/s" if "lin" in k else ""))
