"""rms / max error of the window-attention backward (dq, dk, dv, drel) against fp32 autograd: window kernels vs general kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grove_amd import ops, _lib
bf16 = torch.bfloat16
dev = torch.device("cuda:0")
B, H, L, hs, hd = 6, 16, 196, 96, 80
kh = kw = 14
khp = 16
alpha = hd ** -0.5
g = torch.Generator().manual_seed(3)
qkv = torch.zeros(B * L, 3, H, hs)
qkv[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g)
qkv = qkv.reshape(B * L, 3 * H * hs).to(bf16)
do = torch.zeros(B * L, H, hs)
do[..., :hd] = torch.randn(B * L, H, hd, generator=g)
do = do.reshape(B * L, H * hs).to(bf16)
relp = torch.zeros(B * H, L, 32)
relp[..., :kh] = torch.randn(B * H, L, kh, generator=g) / alpha
relp[..., khp:khp + kw] = torch.randn(B * H, L, kw, generator=g) / alpha
relp = relp.to(bf16)
t = qkv.double().view(B, L, 3, H, hs).requires_grad_(True)
q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)
relr = relp.double().clone().requires_grad_(True)
s = q @ k.transpose(-1, -2) * alpha + ((relr[..., :kh, None] + relr[..., None, khp:khp + kw]) * alpha).reshape(B, H, L, L)
o_ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H * hs)
o_ref.backward(do.double())
gref = t.grad.reshape(B * L, 3 * H * hs).float()
for arm in (1, 0):
    _lib.lib().grove_flash_attn_set_window_kernels(arm)
    out, lse = ops.flash_attn(qkv.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(khp, kw), want_lse=True, hs_valid=hd)
    dqkv = torch.zeros_like(qkv, device=dev)
    drel = ops.flash_attn_bwd(qkv.to(dev), out, do.to(dev), lse, dqkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(khp, kw),
                              want_drel=True, hs_valid=hd)
    res = {"out": ((out.float().cpu() - o_ref.float()).pow(2).mean().sqrt() / o_ref.float().pow(2).mean().sqrt()).item()}
    for name, c0 in (("dq", 0), ("dk", H * hs), ("dv", 2 * H * hs)):
        a, b = dqkv[:, c0:c0 + H * hs].float().cpu(), gref[:, c0:c0 + H * hs]
        res[name] = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
        res[name + "_bias"] = ((a - b).sum() / b.abs().sum()).item()
    a, b = drel.float().cpu(), relr.grad.float()
    res["drel"] = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    print("window kernels" if arm else "general kernels", {k_: round(v_, 5) for k_, v_ in res.items()})
_lib.lib().grove_flash_attn_set_window_kernels(1)
