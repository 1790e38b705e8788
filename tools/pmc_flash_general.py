"""PMC target: the general fused-attention kernels on the step's three non-window shapes (LLaMA causal d=128, SAM global d=80 in
96-wide slots with rel-pos, CLIP d=64), forward + backward, a few launches each.
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... -- python3 tools/pmc_flash_general.py"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
cases = [("sam global", 32, 16, 1024, 96, 80, False, (32, 32)), ("llama", 4, 32, 703, 128, 128, True, None), ("clip", 32, 16, 577, 64, 64, False, None)]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, B, H, L, hs, hd, causal, rel_hw in cases:
    if only and only not in name:
        continue
    qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
    qkv = qkv.to(bf)
    do = torch.randn(B * L, H * hs, device=dev).to(bf)
    rel, arg = None, (0, 0)
    if rel_hw:
        khp = (rel_hw[0] + 15) // 16 * 16
        rel = torch.randn(B * H, L, 2 * khp, device=dev).to(bf)
        arg = (khp, rel_hw[1])
    dq = torch.empty_like(qkv)
    hv = hd if hd < hs else 0
    for _ in range(3):
        out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, causal=causal, rel=rel, rel_hw=arg, want_lse=True, hs_valid=hv)
        ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, causal=causal, rel=rel, rel_hw=arg,
                           want_drel=rel is not None, hs_valid=hv)
torch.cuda.synchronize()
print("done")
