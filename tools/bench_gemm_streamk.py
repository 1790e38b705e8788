"""Stream-K tail of the pipelined GEMMs: on vs off per shape of the full-size step (time, TF/s, max |diff| between the arms and vs fp32)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
shapes = [(32768, 1280, 5120), (32768, 5120, 1280), (32768, 3840, 1280), (32768, 1280, 3840), (32768, 1280, 1280), (2812, 22016, 4096),
          (2812, 4096, 22016), (2812, 12288, 4096), (2812, 4096, 12288), (2812, 11008, 4096), (2812, 4096, 11008), (2812, 4096, 4096),
          (18464, 1024, 4096), (18464, 4096, 1024), (18464, 3072, 1024), (18464, 1024, 1024), (32768, 4608, 1280), (32768, 1280, 4608),
          (1000, 520, 192), (5000, 776, 64), (70000, 512, 128), (777, 1000, 2048), (4000, 1300, 8192)]
tot = {0: 0.0, 1: 0.0}
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    outs, res, S = {}, {0: 1e9, 1: 1e9}, 0
    for rnd_ in range(3):
        for on in (0, 1):
            L.grove_gemm_set_stream_k(on)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.linear(a, b, bias, out=out)
            if on:
                S = L.grove_gemm_last_stream_k()
            var = L.grove_gemm_last_variant()
            torch.cuda.synchronize()
            outs[on] = out
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.linear(a, b, bias, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[on] = min(res[on], e0.elapsed_time(e1) / 10 * 1e3)
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    rows = slice(0, min(M, 2048))
    ref = a[rows].float() @ b.float().t() + bias.float()
    err = ((outs[1][rows].float() - ref).abs().max() / ref.abs().max()).item()
    tot[0] += res[0]
    tot[1] += res[1]
    print(f"M={M} N={N} K={K} variant {var} S={S}: off {res[0]:7.1f}us  on {res[1]:7.1f}us  ({2.0*M*N*K/res[1]/1e6:.0f} TF on, {res[0]/res[1]:.3f}x)  "
          f"|on-off| {d:.3g}  rel err vs fp32 {err:.2e}", flush=True)
L.grove_gemm_set_stream_k(1)
print(f"sum: off {tot[0]:.0f}us on {tot[1]:.0f}us")
