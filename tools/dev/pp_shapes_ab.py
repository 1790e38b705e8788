"""A/B of two library builds on the step's LLaMA / CLIP GEMM shapes (plain epilogues). usage: pp_shapes_ab.py product _s2"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from grove_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
def t(fn):
    best = 1e9
    for _ in range(4):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4 * 1e3)
    return best
for name, M, N, K in [("LLaMA qkv", 2812, 12288, 4096), ("LLaMA qkv dgrad", 2812, 4096, 12288), ("LLaMA gate|up dgrad", 2812, 4096, 22016),
                      ("LLaMA down", 2812, 4096, 11008), ("LLaMA o_proj", 2812, 4096, 4096), ("CLIP qkv", 18464, 3072, 1024), ("CLIP out_proj", 18464, 1024, 1024),
                      ("CLIP fc2", 18464, 1024, 4096), ("SAM neck-ish", 32768, 1280, 3840)]:
    x = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.03).to(bf); out = torch.empty(M, N, device=dev, dtype=bf)
    us = t(lambda: ops.linear(x, w, out=out))
    print("%%-22s (%%5d, %%5d, %%5d): %%7.1f us %%7.1f TF" %% (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
# LLaMA MLP: gate|up with the SwiGLU pair epilogue (+ aux = gate|up saved), and the down-projection dgrad with the SwiGLU backward epilogue
M, H, I = 2812, 4096, 11008
h = torch.randn(M, H, device=dev).to(bf); wgu = ops.swiglu_interleave((torch.randn(2 * I, H, device=dev) * 0.03).to(bf))
gu = torch.empty(M, 2 * I, device=dev, dtype=bf); a = torch.empty(M, I, device=dev, dtype=bf)
us = t(lambda: ops.linear(h, wgu, act=ops.ACT_SWIGLU_PAIR, aux=gu, ld_aux=2 * I, out=a))
print("%%-22s (%%5d, %%5d, %%5d): %%7.1f us %%7.1f TF" %% ("LLaMA gate|up SwiGLU", M, 2 * I, H, us, 2.0 * M * 2 * I * H / us / 1e6))
dx = torch.randn(M, H, device=dev).to(bf); wdt = (torch.randn(I, H, device=dev) * 0.03).to(bf); dgu = torch.empty(M, 2 * I, device=dev, dtype=bf)
us = t(lambda: ops.linear(dx, wdt, act=ops.ACT_SWIGLU_BWD, residual=gu, out=dgu))
print("%%-22s (%%5d, %%5d, %%5d): %%7.1f us %%7.1f TF" %% ("LLaMA down dgrad SwiGLU'", M, I, H, us, 2.0 * M * I * H / us / 1e6))
''' % ROOT
names = sys.argv[1:] or ["product", "_s2"]
for rnd in range(2):
    for name in names:
        name = "" if name == "product" else name
        print("==", name or "product", flush=True)
        subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, GROVE_HIP_LIB=os.path.join(ROOT, "grove_amd", "csrc", "libgrove_hip%s.so" % name)))
