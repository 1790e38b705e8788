"""Persistent NT GEMM: 192-row against 256-row tiles per shape (grove_gemm_set_tile_m(193 / 256) forces the instance; 0 = the cost model)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0"); bf = torch.bfloat16
L = _lib.lib()
def t(fn):
    best = 1e9
    for _ in range(4):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4 * 1e3)
    return best
M, H, I = 2812, 4096, 11008
cases = []
h = torch.randn(M, H, device=dev).to(bf); wgu = ops.swiglu_interleave((torch.randn(2 * I, H, device=dev) * 0.03).to(bf))
gu = torch.empty(M, 2 * I, device=dev, dtype=bf); a = torch.empty(M, I, device=dev, dtype=bf)
cases.append(("LLaMA gate|up SwiGLU (2812, 22016, 4096)", lambda: ops.linear(h, wgu, act=ops.ACT_SWIGLU_PAIR, aux=gu, ld_aux=2 * I, out=a)))
dx = torch.randn(M, H, device=dev).to(bf); wdt = (torch.randn(I, H, device=dev) * 0.03).to(bf); dgu = torch.empty(M, 2 * I, device=dev, dtype=bf)
cases.append(("LLaMA down dgrad SwiGLU' (2812, 11008, 4096)", lambda: ops.linear(dx, wdt, act=ops.ACT_SWIGLU_BWD, residual=gu, out=dgu)))
for name, m, n, k in [("LLaMA qkv", 2812, 12288, 4096), ("LLaMA gate|up dgrad", 2812, 4096, 22016), ("LLaMA down", 2812, 4096, 11008), ("LLaMA o_proj", 2812, 4096, 4096),
                      ("CLIP qkv", 18464, 3072, 1024), ("CLIP out_proj", 18464, 1024, 1024), ("CLIP fc2", 18464, 1024, 4096), ("SAM qkv", 32768, 3840, 1280),
                      ("SAM proj", 32768, 1280, 1280), ("SAM fc2", 32768, 1280, 5120), ("SAM fc2 dgrad", 32768, 5120, 1280)]:
    x = torch.randn(m, k, device=dev).to(bf); w = (torch.randn(n, k, device=dev) * 0.03).to(bf); out = torch.empty(m, n, device=dev, dtype=bf)
    cases.append((f"{name} ({m}, {n}, {k})", (lambda x=x, w=w, out=out: ops.linear(x, w, out=out))))
for name, fn in cases:
    r = {}
    for mode in (0, 193, 256):
        L.grove_gemm_set_tile_m(mode)
        r[mode] = t(fn)
    L.grove_gemm_set_tile_m(0)
    pick = "192" if abs(r[0] - r[193]) < abs(r[0] - r[256]) else "256"
    print(f"{name:48s} model {r[0]:7.1f} us (~{pick}) | 192-row {r[193]:7.1f} | 256-row {r[256]:7.1f}", flush=True)
