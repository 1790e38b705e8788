import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grove_amd import ops, _lib
bf16 = torch.bfloat16
dev = torch.device("cuda:0")
B, H, L, hs, hd = 2, 2, 196, 96, 80
kh = kw = 14
khp = 16
alpha = hd ** -0.5

def run(tag, use_rel=True, dmax=80, vmode="rand", qk_scale=1.0):
    g = torch.Generator().manual_seed(1)
    qkv = torch.zeros(B * L, 3, H, hs)
    qkv[..., :dmax] = torch.randn(B * L, 3, H, dmax, generator=g)
    qkv[:, :2] *= qk_scale
    if vmode == "ones":
        qkv[:, 2] = 0
        qkv[:, 2, :, :hd] = 1.0
    if vmode == "rowid":
        qkv[:, 2] = 0
        qkv[:, 2, :, :hd] = (torch.arange(B * L) % L).float()[:, None, None] / 64
    qkv = qkv.reshape(B * L, 3 * H * hs).to(bf16)
    relp = torch.zeros(B * H, L, 32)
    if use_rel:
        relp[..., :kh] = torch.randn(B * H, L, kh, generator=g) / alpha
        relp[..., khp:khp + kw] = torch.randn(B * H, L, kw, generator=g) / alpha
    relp = relp.to(bf16)
    t = qkv.float().view(B, L, 3, H, hs)
    q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) * alpha
    bias = (relp.float()[..., :kh, None] + relp.float()[..., None, khp:khp + kw]) * alpha
    s = s + bias.reshape(B, H, L, L)
    p = torch.softmax(s, -1)
    o_ref = (p @ v).transpose(1, 2).reshape(B * L, H * hs)
    lse_ref = torch.logsumexp(s, -1).reshape(B * H, L)
    out, lse = ops.flash_attn(qkv.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(khp, kw), want_lse=True, hs_valid=hd)
    err = (out.float().cpu() - o_ref).abs()
    e2 = err.view(B, L, H, hs)
    print(f"{tag:28s} out err max {err.max():.4f} (ref max {o_ref.abs().max():.3f})  lse err {(lse.cpu() - lse_ref).abs().max():.4f}"
          f"  worst q {int(e2.amax((0, 2, 3)).argmax())} worst d {int(e2.amax((0, 1, 2)).argmax())}  err by q-tile {[round(float(e2[:, i*16:(i+1)*16].max()), 3) for i in range(13)]}")

run("rel=0 V=ones qk=0", use_rel=False, vmode="ones", qk_scale=0.0)
run("rel=0 V=rowid qk=0", use_rel=False, vmode="rowid", qk_scale=0.0)
run("rel=0 V=rand qk=0", use_rel=False, qk_scale=0.0)
run("rel=0 d<64", use_rel=False, dmax=64)
run("rel=0 d<80", use_rel=False)
run("rel only (qk=0)", use_rel=True, qk_scale=0.0)
run("full", use_rel=True)
