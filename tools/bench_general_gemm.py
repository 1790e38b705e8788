"""The general (non-persistent) NT and TN GEMM kernels on the step's small shapes: decoder-sized NT products and the trainable layers'
weight gradients. Median of interleaved repeats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
def timed(f, reps=50):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in [(576, 256, 256), (576, 2048, 256), (576, 256, 2048), (256, 32008, 4096), (18432, 1024, 608), (2304, 4096, 1024)]:
    a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf)
    print(f"NT ({M}, {N}, {K}): {timed(lambda: ops.linear(a, w)):7.1f} us", flush=True)
for (K, M, N) in [(2304, 4096, 4096), (2304, 4096, 1024), (256, 32008, 4096), (576, 256, 256), (32768, 1280, 1280), (32008, 4096, 256)]:
    dy = torch.randn(K, M, device=dev).to(bf); x = torch.randn(K, N, device=dev).to(bf)
    g = torch.zeros(M, N, dtype=torch.float32, device=dev)
    print(f"TN wgrad K={K} ({M}, {N}): {timed(lambda: ops.wgrad(dy, x, g)):7.1f} us", flush=True)
# the lm_head dgrad in its TN form: split-K sweep
K, M, N = 32008, 4096, 256
dy = torch.randn(K, M, device=dev).to(bf); x = torch.randn(K, N, device=dev).to(bf)
g = torch.zeros(M, N, dtype=torch.float32, device=dev)
for sk in (0, 2, 3, 4, 6, 8, 12):
    print(f"TN lm_head dgrad split_k={sk}: {timed(lambda: ops.wgrad(dy, x, g, split_k=sk)):7.1f} us", flush=True)
