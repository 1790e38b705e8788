"""A/B of the weight-load request shape of the matrix-core GEMV (grove_gemv_set_mfma bit 3) at 8 sequences on LLaMA-7B's projection shapes:
results must be bit-identical (the same products in the same order), only the request shape of the loads differs."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32008, 4096)]:
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
    x = torch.randn(8, K, device=dev).to(torch.bfloat16)
    out = torch.empty(8, N, device=dev, dtype=torch.bfloat16)
    ref = None
    for knob in (9, 1, 9, 1):
        L.grove_gemv_set_mfma(knob)
        for w in ws:
            ops.gemv(x, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for w in ws:
                ops.gemv(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        o = out.clone()
        if ref is None:
            ref = o
        print(f"N={N} K={K} {'operand-shape loads' if knob == 9 else '128 B / row + exchange'}: {us:6.1f} us  {N * K * 2 / us / 1e3:7.1f} GB/s  bit-identical to the first: {bool(torch.equal(o, ref))}", flush=True)
L.grove_gemv_set_mfma(1)
