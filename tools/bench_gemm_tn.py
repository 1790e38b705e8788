"""Plain (un-gathered) TN GEMM, one full round of 256 x 256 tiles: pipelined vs 128 x 128 kernel."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for (K2, M2, N2) in [(8192, 4096, 4096), (16384, 2560, 5120), (32768, 1280, 34560)]:
    a = torch.randn(K2, M2, device=dev).to(torch.bfloat16); b = torch.randn(K2, N2, device=dev).to(torch.bfloat16)
    o = torch.zeros(M2, N2, dtype=torch.float32, device=dev)
    for mode in (0, 1):
        L.grove_gemm_tn_set_pipelined(mode)
        best = 1e9
        for _ in range(3):
            ops.wgrad(a, b, o); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.wgrad(a, b, o); ops.wgrad(a, b, o); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 2 * 1e3)
        print(f"plain wgrad K={K2} M={M2} N={N2} mode={mode}: {best:8.1f} us  {2.0*M2*N2*K2/best/1e6:7.1f} TF", flush=True)
L.grove_gemm_tn_set_pipelined(-1)
# the partial last round cut into K ranges (grove_gemm_tn_set_split_tail): the SAM adapter Conv3d weight gradient, plain and gathered
from grove_amd.model.indexing import conv3d_gather_index
for gathered in (False, True):
    K2, M2, Ci = 32768, 1280, 1280
    a = torch.randn(K2, M2, device=dev).to(torch.bfloat16)
    b = torch.randn(K2, Ci if gathered else 27 * Ci, device=dev).to(torch.bfloat16)
    idx = conv3d_gather_index(2, 16, 32, 32).to(dev) if gathered else None
    o = torch.zeros(M2, 27 * Ci, dtype=torch.float32, device=dev)
    for on in (0, 2, 0, 2):
        L.grove_gemm_tn_set_split_tail(on)
        kw = dict(b_idx=idx, b_taps=27) if gathered else {}
        best = 1e9
        for _ in range(3):
            ops.wgrad(a, b, o, **kw); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.wgrad(a, b, o, **kw); ops.wgrad(a, b, o, **kw); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 2 * 1e3)
        print(f"{'gathered' if gathered else 'plain   '} wgrad (1280, 34560, 32768) split_tail={on} parts={L.grove_gemm_tn_last_parts()}: {best:8.1f} us  "
              f"{2.0 * M2 * 27 * Ci * K2 / best / 1e6:7.1f} TF", flush=True)
L.grove_gemm_tn_set_split_tail(1)
