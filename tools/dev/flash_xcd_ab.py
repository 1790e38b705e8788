"""Same-box A/B of the XCD-grouped launch order of the non-causal eight-wave attention kernels (grove_flash_attn_set_v2 bit 5):
SAM global blocks (32 x 32 tokens, head dim 96, rel-pos) and CLIP (577 tokens, head dim 64) at training and inference batch sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grove_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def run(name, B, H, L, hs, hd, rel_hw, bwd=True):
    alpha = hd ** -0.5
    qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
    qkv = qkv.to(bf)
    rel = (torch.randn(B * H, L, 64, device=dev) * 0.3).to(bf) if rel_hw else None
    do = torch.randn(B * L, H * hs, device=dev).to(bf)
    dq = torch.empty_like(qkv) if bwd else None
    kw = dict(rel=rel, rel_hw=rel_hw or (0, 0), hs_valid=hd if hd < hs else 0)
    res = {}
    for mask in (15, 47, 15, 47):
        _lib.lib().grove_flash_attn_set_v2(mask)
        out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, want_lse=True, **kw)
        fns = [lambda: ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, want_lse=True, out=out, **kw)]
        if bwd:
            fns.append(lambda: ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, want_drel=rel is not None, **kw))
        ts = []
        for fn in fns:
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        res.setdefault(mask, []).append((ts, out.float().sum().item(), dq.float().sum().item() if bwd else 0.0))
    for mask, runs in res.items():
        print(f"{name:28s} mask {mask}: " + " | ".join("fwd %.0f us" % r[0][0] + (" bwd %.0f us" % r[0][1] if bwd else "") for r in runs), flush=True)
    a, b = res[15][0], res[47][0]
    print(f"{'':28s} checksums equal: {a[1] == b[1] and a[2] == b[2]}", flush=True)


run("sam global 32 frames", 32, 16, 1024, 96, 80, (32, 32))
run("sam global 320 frames", 320, 16, 1024, 96, 80, (32, 32), bwd=False)
run("clip 32 frames", 32, 16, 577, 64, 64, None)
run("clip 320 frames", 320, 16, 577, 64, 64, None, bwd=False)
run("llama-like non-causal 4x703", 4, 32, 703, 128, 128, None)
