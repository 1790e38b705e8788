"""Gathered-A GEMMs of the step (Conv3d adapter, window partition) on the pipelined kernel vs the two-barrier kernel."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
from grove_amd.model.indexing import conv3d_gather_index, window_partition_index
dev = torch.device("cuda:0")
L = _lib.lib()
def bench(fn):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3 * 1e3)
    return best
# SAM adapter conv: 4 groups x 8 frames x 32 x 32, 1280 -> 1280
idx = conv3d_gather_index(4, 8, 32, 32).to(dev)
M = 32768
x = torch.randn(M, 1280, device=dev).to(torch.bfloat16)
w = (torch.randn(1280, 27 * 1280, device=dev) * 0.02).to(torch.bfloat16)
b = torch.randn(1280, device=dev).to(torch.bfloat16)
sc = torch.tensor([0.3], device=dev)
for tm in (128, 0):
    L.grove_gemm_set_tile_m(tm)
    t = bench(lambda: ops.linear(x, w, b, act=ops.ACT_RELU, residual=x, scale_ptr=sc, scale_tanh=True, a_idx=idx, a_taps=27, M=M))
    print(f"conv3d 32768x1280x34560 tile_m={tm}: {t:8.1f} us  {2.0*M*1280*34560/t/1e6:7.1f} TF  variant {L.grove_gemm_last_variant()}", flush=True)
# window gather: tokens 32768 <- padded windows 56448
tok2win, win2tok, nwin, _, _ = window_partition_index(32, 32, 32, 14)
tok2win = tok2win.to(dev)
o = torch.randn(56448, 1536, device=dev).to(torch.bfloat16)
wp = (torch.randn(1280, 1536, device=dev) * 0.02).to(torch.bfloat16)
for tm in (128, 0):
    L.grove_gemm_set_tile_m(tm)
    t = bench(lambda: ops.linear(o, wp, b, residual=x, a_idx=tok2win, a_taps=1, M=M))
    print(f"proj gather 32768x1280x1536 tile_m={tm}: {t:8.1f} us  {2.0*M*1280*1536/t/1e6:7.1f} TF  variant {L.grove_gemm_last_variant()}", flush=True)
L.grove_gemm_set_tile_m(0)
