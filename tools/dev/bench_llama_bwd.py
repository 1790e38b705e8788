"""Round-5 debug: LLaMA-shaped attention backward with the fused inverse RoPE, old vs new kernels (grove_flash_attn_set_v2 masks)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
bf = torch.bfloat16
B, H, L, hs = 4, 32, 703, 128
qkv = torch.randn(B * L, 3 * H * hs, device=dev).to(bf)
do = torch.randn(B * L, H * hs, device=dev).to(bf)
alpha = hs ** -0.5
table = ops.rope_table(hs, 10000.0, L + 8, dev)
out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=True, want_lse=True)
dq = torch.empty_like(qkv)
res = {}
for mask in (0, 1 | 2, 1 | 2 | 4 | 8, 1 | 4 | 8):
    _lib.lib().grove_flash_attn_set_v2(mask)
    for rope in (None, table):
        f = lambda: ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=True, rope=rope)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        res[(mask, rope is not None)] = (t, dq.float().clone())
        print(f"v2 mask {mask:2d} rope {rope is not None}: {t:7.1f} us per backward (dq + dkv)", flush=True)
_lib.lib().grove_flash_attn_set_v2(47)
ref = res[(0, True)][1]
for k, (t, g) in res.items():
    if k[1]:
        d = (g - ref).abs()
        print(k, "max |diff vs old kernels|", float(d.max()), "mean", float(d.mean()))
