"""The box decoder's long-K weight gradients (dW[M, N] += dz[K, M]^T x[K, N], K = instances x 1024 image tokens, tiny M x N): time of
grove_gemm_tn_bf16 over its K-split count."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grove_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
for (M, N, K) in ((128, 256, 98304), (256, 128, 98304), (256, 256, 98304), (256, 256, 576), (2048, 256, 576), (256, 2048, 576), (128, 256, 576)):
    dy = torch.randn(K, M, device=dev).to(bf)
    x = torch.randn(K, N, device=dev).to(bf)
    g = torch.zeros(M, N, device=dev)
    line = []
    for split in (0, 4, 8, 16, 32, 64, 128, 256):
        if split > max(K // 128, 1):
            continue
        for _ in range(2):
            ops.wgrad(dy, x, g, K=K, split_k=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.wgrad(dy, x, g, K=K, split_k=split)
        e1.record(); torch.cuda.synchronize()
        line.append(f"split {split}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    print(f"M={M} N={N} K={K}: " + " | ".join(line), flush=True)
