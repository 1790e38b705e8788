"""PMC target: SAM window attention fwd + bwd (a few launches)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
B, H, L, hs, hd = 288, 16, 196, 96, 80
qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
qkv = qkv.to(bf)
do = torch.randn(B * L, H * hs, device=dev).to(bf)
rel = torch.randn(B * H, L, 32, device=dev).to(bf)
dq = torch.empty_like(qkv)
from grove_amd import _lib
for window_kernels in (0, 1):  # 0: the general flash kernels on the window shape (round 1), 1: the LDS-resident window kernels
    _lib.lib().grove_flash_attn_set_window_kernels(window_kernels)
    for _ in range(3):
        out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, rel=rel, rel_hw=(16, 14), want_lse=True, hs_valid=hd)
        ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, rel=rel, rel_hw=(16, 14), want_drel=True,
                           hs_valid=hd)
# round 6b: the same problem with the rel-pos terms made inside the window kernels (grove_flash_attn_params.rel_table): win_attn_*_kernel<true>
_lib.lib().grove_flash_attn_set_window_kernels(1)
T = ops.rel_table_images((torch.randn(27, hd, device=dev) * 0.3).to(bf), (torch.randn(27, hd, device=dev) * 0.3).to(bf), 14, hd ** -0.5)
rel_o = torch.empty((B * H, L, 32), dtype=bf, device=dev)
for _ in range(3):
    out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, rel_hw=(16, 14), want_lse=True, hs_valid=hd, rel_table=T, rel_out=rel_o)
    ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, rel=rel_o, rel_hw=(16, 14), hs_valid=hd, rel_table=T)
torch.cuda.synchronize()
print("done")
