"""Round-5 debug: dump S^T of one tile from the eight-wave forward (debug library) and compare with torch."""
import os, sys, ctypes as C
os.environ["GROVE_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "grove_amd", "csrc", "libgrove_hip_dbg.so")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grove_amd import ops, _lib
from grove_amd.ops import _p, _stream
dev = torch.device("cuda:0")
bf = torch.bfloat16

def probe(B, H, L, hs, hd, stage):
    g = torch.Generator().manual_seed(1)
    qkv = torch.zeros(B * L, 3 * H * hs)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g)
    qkv = qkv.to(bf).to(dev)
    alpha = hd ** -0.5
    out = torch.zeros(B * L, H * hs, dtype=bf, device=dev)
    dump = torch.full((B * H, L, 64), float("nan"), dtype=torch.float32, device=dev)
    ld = qkv.stride(0)
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o = _p(qkv[:, 0:]), _p(qkv[:, H * hs:]), _p(qkv[:, 2 * H * hs:]), _p(out)
    p.delta = _p(dump)
    p.d_o = stage
    p.sq = p.sk = p.sv = L * ld
    p.so = L * out.stride(0)
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, L, L, hs
    p.ld_q = p.ld_k = p.ld_v = ld
    p.ld_o = out.stride(0)
    p.alpha = alpha
    _lib.check(_lib.lib().grove_flash_attn_fwd(C.byref(p), _stream()), "fwd")
    torch.cuda.synchronize()
    t = qkv.float().view(B, L, 3, H, hs)
    q, k = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2)
    sc = alpha * 1.4426950408889634
    qs = (q * sc).to(bf).float()
    nt = (L + 63) // 64
    k0 = 0 if stage == 1 else (nt - 1) * 64
    idx = torch.arange(k0, k0 + 64, device=dev).clamp(max=L - 1)
    ref = (qs @ k[:, :, idx].transpose(-1, -2)).reshape(B * H, L, 64)
    d = (dump - ref).abs()
    bad = torch.isnan(d) | (d > 0.05)
    print(f"B{B} H{H} L{L} hs{hs} stage{stage}: max err {float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else -1:.4f}  nan {int(torch.isnan(dump).sum())}  bad {int(bad.sum())} / {bad.numel()}")
    if bad.any():
        bh, qq, kk = bad.nonzero()[:1][0].tolist()
        print("  first bad (bh, q, key):", bh, qq, kk, " got", dump[bh, qq, :8].tolist(), "\n   want", ref[bh, qq, :8].tolist())
        # which keys / queries are bad
        print("  bad per key col (first bh):", bad[0].any(0).int().tolist())
        print("  bad per q row (first bh, first 64):", bad[0].any(1).int()[:64].tolist())
        # does got match some OTHER key? permutation hunt on row 0
        row = dump[0, 0]
        full = (qs[0, 0, 0:1] @ k[0, 0].transpose(-1, -2)).flatten()
        match = [(int((full - v).abs().argmin()), float((full - v).abs().min())) for v in row[:16]]
        print("  row 0 of bh 0, keys 0..15 look like keys:", match)

for hs, hd in ((128, 128), (64, 64), (96, 80)):
    probe(1, 1, 64, hs, hd, 1)
    probe(1, 2, 300, hs, hd, 1)
    probe(1, 2, 300, hs, hd, 2)
