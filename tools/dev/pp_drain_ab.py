"""A/B of the persistent NT GEMM with / without a compiler-visible drain after the epilogue (libgrove_hip_drain{all,sel}.so)."""
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
cases = [("SAM fc1 + GELU + aux", 32768, 5120, 1280, ops.ACT_GELU, True, False), ("SAM fc1 + GELU", 32768, 5120, 1280, ops.ACT_GELU, False, False),
         ("CLIP fc1 + QuickGELU", 18464, 4096, 1024, ops.ACT_QUICKGELU, False, False), ("SAM fc2 + residual", 32768, 1280, 5120, ops.ACT_NONE, False, True),
         ("SAM qkv", 32768, 3840, 1280, ops.ACT_NONE, False, False), ("SAM proj + residual", 32768, 1280, 1280, ops.ACT_NONE, False, True),
         ("SAM fc2 dgrad", 32768, 5120, 1280, ops.ACT_NONE, False, False), ("LLaMA o_proj + residual", 2812, 4096, 4096, ops.ACT_NONE, False, True)]
for name, M, N, K, act, aux, res in cases:
    x = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.03).to(bf); b = torch.randn(N, device=dev).to(bf)
    r = torch.randn(M, N, device=dev).to(bf) if res else None
    a = torch.empty(M, N, device=dev, dtype=bf) if aux else None
    out = torch.empty(M, N, device=dev, dtype=bf)
    us = t(lambda: ops.linear(x, w, b, act=act, residual=r, aux=a, out=out))
    print("%%-26s (%%5d, %%5d, %%5d): %%7.1f us %%7.1f TF" %% (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
from grove_amd.model.indexing import conv3d_gather_index
G, T, H, W, C = 4, 8, 32, 32, 1280
M = G * T * H * W
x = torch.randn(M, C, device=dev).to(bf); w = (torch.randn(C, 27 * C, device=dev) * 0.02).to(bf); b = torch.randn(C, device=dev).to(bf)
idx = conv3d_gather_index(G, T, H, W).to(dev); alpha = torch.full((1,), 0.3, device=dev)
pre = torch.empty(M, C, device=dev, dtype=bf); out = torch.empty(M, C, device=dev, dtype=bf)
us = t(lambda: ops.linear(x, w, b, act=ops.ACT_RELU, scale_ptr=alpha, scale_tanh=True, a_idx=idx, a_taps=27, M=M, residual=x, aux=pre, a_frames=(H * W, T), out=out))
print("%%-26s (%%5d, %%5d, %%5d): %%7.1f us %%7.1f TF" %% ("adapter fwd <gather, 1>", M, C, 27 * C, us, 2.0 * M * C * 27 * C / us / 1e6))
us = t(lambda: ops.linear(x, w, a_idx=idx, a_taps=27, M=M, residual=x, scale_ptr=alpha, scale_tanh=True, a_frames=(H * W, T), out=out))
print("%%-26s (%%5d, %%5d, %%5d): %%7.1f us %%7.1f TF" %% ("adapter dgrad <gather, 0>", M, C, 27 * C, us, 2.0 * M * C * 27 * C / us / 1e6))
''' % ROOT
names = sys.argv[1:] or ["", "_drainall", "_drainsel"]
for rnd in range(2):
    for name in names:
        name = "" if name == "product" else name
        lib = os.path.join(ROOT, "grove_amd", "csrc", "libgrove_hip%s.so" % name)
        print("==", name or "product", flush=True)
        subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, GROVE_HIP_LIB=lib))
