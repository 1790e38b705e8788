"""Auto tile selection vs every forced variant on the GEMM shapes of the full-size step (bias epilogue)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
shapes = [(32768, 5120, 1280), (32768, 1280, 5120), (56448, 4608, 1280), (2812, 22016, 4096), (2812, 4096, 22016), (56448, 1280, 4608),
          (2812, 12288, 4096), (2812, 11008, 4096), (56448, 1280, 1536), (2812, 4096, 12288), (2812, 4096, 11008), (18464, 4096, 1024),
          (2812, 4096, 4096), (56448, 1536, 1280), (18464, 1024, 4096), (18464, 3072, 1024), (18464, 1024, 1024), (32768, 4608, 1280),
          (2304, 4096, 4096), (256, 32008, 4096), (98304, 256, 128), (576, 256, 256), (1000, 520, 192), (777, 1000, 128), (300, 264, 64), (5000, 776, 64), (70000, 512, 128)]
variants = [("auto", 0), ("192", 192), ("pp192", 193), ("pp256", 256)]
tot = {v[0]: 0.0 for v in variants}
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {v[0]: 1e9 for v in variants}
    for rnd_ in range(3):
        for name, tm in variants:
            L.grove_gemm_set_tile_m(tm)
            ops.linear(a, b, bias, out=out)
            torch.cuda.synchronize()
            if rnd_ == 0:
                if name == "auto":
                    ref = out.clone()
                else:
                    # bit-identical unless a stream-K split changed the fp32 sum order of some tiles (then: bf16 rounding)
                    err = (out.float() - ref.float()).abs().max().item()
                    assert err <= 2 ** -7 * ref.float().abs().max().item(), (name, M, N, K, err)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(a, b, bias, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[name] = min(res[name], e0.elapsed_time(e1) / 5 * 1e3)
    best = min(res[k] for k in ("192", "pp192", "pp256"))
    flag = "" if res["auto"] <= 1.04 * best else "   <-- auto misses"
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k}: {v:7.1f}us" for k, v in res.items()) + f"  ({2.0*M*N*K/res['auto']/1e6:.0f} TF auto)" + flag, flush=True)
L.grove_gemm_set_tile_m(0)
