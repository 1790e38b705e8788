"""Is the 256x256 epilogue chip-bandwidth-bound or per-CU issue-bound? One-round launches at full / half / quarter chip, two K values -> fixed cost c."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
L.grove_gemm_set_tile_m(256)
N = 2560
def t(M, K, f32=False):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    best = 1e9
    for _ in range(5):
        ops.gemm_raw(a, b, out, M, N, K, K, K, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_raw(a, b, out, M, N, K, K, K, N)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return best
for f32 in (False, True):
    for M in (6400, 3328, 1792, 12800):
        t1, t2 = t(M, 256, f32), t(M, 1280, f32)
        k = (t2 - t1) / 16
        print(f"f32={f32} M={M} tiles={(M+255)//256*10}: t(K=256)={t1:.1f}us t(K=1280)={t2:.1f}us  per-K-tile={k:.2f}us fixed={t1-4*k:.1f}us", flush=True)
L.grove_gemm_set_tile_m(0)
