"""What do the row scatter (c_idx) and the padded-head column map (n_map) of SAM's window qkv GEMM cost? (32768, 3840, 1280) with bias:
plain / c_idx only / n_map only / both (the product's launch). Interleaved repeats, median."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
M, N, K, rows_w = 32768, 3840, 1280, 56448
a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev).to(bf)
perm = torch.randperm(rows_w, device=dev)[:M].to(torch.int32).sort().values  # token -> windowed row (monotone, like window_partition)
cases = {
    "plain": dict(),
    "c_idx": dict(c_idx=perm, out_rows=rows_w),
    "n_map": dict(out_cols=4608, n_map=(80, 16)),
    "c_idx + n_map": dict(c_idx=perm, out_rows=rows_w, out_cols=4608, n_map=(80, 16)),
}
def timed(f, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {k: [] for k in cases}
for k, kw in cases.items(): ops.linear(a, w, b, **kw)
for rep in range(5):
    for k, kw in cases.items():
        res[k].append(timed(lambda: ops.linear(a, w, b, **kw)))
for k, v in res.items():
    us = sorted(v)[len(v) // 2]
    print(f"{k:14s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
