"""Backward of the general attention kernels at head dim 128 over sequence lengths (non-causal, B*H = 256 = one key block of 128 per CU at
L = 128): separates the per-block fixed cost from the cost per 64-query tile."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
B, H, hs = 8, 32, 128
for L in (128, 256, 512, 1024, 2048):
    qkv = torch.randn(B * L, 3 * H * hs, device=dev).to(bf)
    do = torch.randn(B * L, H * hs, device=dev).to(bf)
    dq = torch.empty_like(qkv)
    out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, hs ** -0.5, want_lse=True)
    def bwd():
        ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, hs ** -0.5)
    bwd(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): bwd()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    kb = (L + 127) // 128
    print(f"L={L:5d}: bwd {ms * 1e3:8.1f} us  key blocks per (b,h) {kb}, 64-query tiles per block {L // 64}, block-tiles per CU {B * H * kb * (L // 64) / 256:.0f}, {10.0 * B * H * L * L * hs / ms / 1e9:6.1f} TF/s", flush=True)
