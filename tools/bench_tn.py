"""TN weight-gradient GEMM: the persistent pipelined kernel vs the 128 x 128 kernel on the SAM adapter Conv3d shape."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
from grove_amd.model.indexing import conv3d_gather_index
dev = torch.device("cuda:0")
L = _lib.lib()
idx = conv3d_gather_index(4, 8, 32, 32).to(dev)
K, M, Ci = 32768, 1280, 1280
dz = torch.randn(K, M, device=dev).to(torch.bfloat16)
x = torch.randn(K, Ci, device=dev).to(torch.bfloat16)
out = torch.zeros(M, 27 * Ci, dtype=torch.float32, device=dev)
for mode in (0, 1, -1):
    L.grove_gemm_tn_set_pipelined(mode)
    best = 1e9
    for _ in range(3):
        ops.wgrad(dz, x, out, b_idx=idx, b_taps=27); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.wgrad(dz, x, out, b_idx=idx, b_taps=27); ops.wgrad(dz, x, out, b_idx=idx, b_taps=27); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 2 * 1e3)
    print(f"conv3d wgrad M={M} N={27*Ci} K={K} mode={mode}: {best:8.1f} us  {2.0*M*27*Ci*K/best/1e6:7.1f} TF", flush=True)
# plain TN: mm_projector-like / lm_head-like
for (K2, M2, N2) in [(2812, 4096, 4096), (2304, 4096, 1024), (32768, 1280, 1280)]:
    a = torch.randn(K2, M2, device=dev).to(torch.bfloat16); b = torch.randn(K2, N2, device=dev).to(torch.bfloat16)
    o = torch.zeros(M2, N2, dtype=torch.float32, device=dev)
    for mode in (0, 1, -1):
        L.grove_gemm_tn_set_pipelined(mode)
        best = 1e9
        for _ in range(3):
            ops.wgrad(a, b, o); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.wgrad(a, b, o); ops.wgrad(a, b, o); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 2 * 1e3)
        print(f"wgrad K={K2} M={M2} N={N2} mode={mode}: {best:8.1f} us  {2.0*M2*N2*K2/best/1e6:7.1f} TF", flush=True)
L.grove_gemm_tn_set_pipelined(-1)
