"""Round-5 debug: ablations of the eight-wave forward (debug library; wrong results by construction, right instruction streams)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GROVE_HIP_LIB"] = os.path.join(ROOT, "grove_amd", "csrc", "libgrove_hip_dbg.so")
import torch
sys.path.insert(0, ROOT)
from grove_amd import ops, _lib
from grove_amd.ops import _p, _stream
dev = torch.device("cuda:0")
bf = torch.bfloat16
NAMES = {0: "as shipped", 1: "no DMA in the loop", 2: "no softmax", 3: "no DMA, no softmax", 4: "no X (MFMAs + reads)", 5: "no DMA, no X", 6: "no softmax, no X", 7: "barriers only"}

def run(name, B, H, L, hs, hd, causal, rel_hw):
    qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
    qkv = qkv.to(bf)
    alpha = hd ** -0.5
    out = torch.zeros(B * L, H * hs, dtype=bf, device=dev)
    rel = (torch.randn(B * H, L, 64, device=dev) / alpha).to(bf) if rel_hw else None
    dummy = torch.zeros(8 * 32 * 8, dtype=torch.int64, device=dev)
    ld = qkv.stride(0)
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o = _p(qkv[:, 0:]), _p(qkv[:, H * hs:]), _p(qkv[:, 2 * H * hs:]), _p(out)
    p.delta = _p(dummy)
    p.rel = _p(rel)
    p.sq = p.sk = p.sv = L * ld
    p.so = L * out.stride(0)
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, L, L, hs
    p.ld_q = p.ld_k = p.ld_v = ld
    p.ld_o = out.stride(0)
    p.alpha = alpha
    p.causal = int(causal)
    if rel_hw:
        p.rel_kh, p.rel_kw, p.rel_ld = 32, 32, 64
    res = []
    for mask in range(8):
        p.d_o = 16 + mask
        f = lambda: _lib.check(_lib.lib().grove_flash_attn_fwd(C.byref(p), _stream()), "fwd")
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"== {name}: " + " | ".join(f"{NAMES[m]} {v:.1f} us" for m, v in enumerate(res)), flush=True)

run("sam global", 32, 16, 1024, 96, 80, False, (32, 32))
run("llama", 4, 32, 703, 128, 128, True, None)
run("clip", 32, 16, 577, 64, 64, False, None)
run("hs128 L2048", 4, 32, 2048, 128, 128, False, None)
