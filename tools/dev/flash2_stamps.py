"""Round-5 debug: s_memtime stamps of the eight-wave forward's segments (debug library), block 0, all 8 waves."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GROVE_HIP_LIB"] = os.path.join(ROOT, "grove_amd", "csrc", "libgrove_hip_dbg.so")
import torch
sys.path.insert(0, ROOT)
from grove_amd import ops, _lib
from grove_amd.ops import _p, _stream
dev = torch.device("cuda:0")
bf = torch.bfloat16

def run(name, B, H, L, hs, hd, causal, rel_hw):
    qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
    qkv = qkv.to(bf)
    alpha = hd ** -0.5
    out = torch.zeros(B * L, H * hs, dtype=bf, device=dev)
    rel = None
    if rel_hw:
        rel = (torch.randn(B * H, L, 64, device=dev) / alpha).to(bf)
    stamps = torch.zeros(8 * 32 * 8 + 64 * 8, dtype=torch.int64, device=dev)
    ld = qkv.stride(0)
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o = _p(qkv[:, 0:]), _p(qkv[:, H * hs:]), _p(qkv[:, 2 * H * hs:]), _p(out)
    p.delta = _p(stamps)
    p.d_o = 3
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
    for _ in range(3):
        _lib.check(_lib.lib().grove_flash_attn_fwd(C.byref(p), _stream()), "fwd")
    torch.cuda.synchronize()
    bs = stamps[8 * 32 * 8:].view(64, 8).cpu()
    st = stamps[:8 * 32 * 8].view(8, 32, 8).cpu()
    nt = (L + 63) // 64
    print(f"== {name}: nt = {nt}; cycles per segment, mean over tiles 2..{min(nt - 2, 31)} (s_memtime ticks)")
    for i in range(32):
        r = bs[i]
        if r[0] == 0:
            continue
        print(f"  item-round {i // 8} block {i % 8}: prologue issue {int(r[1]-r[0])}  wait+sync {int(r[2]-r[1])}  loop {int(r[3]-r[2])}  epilogue {int(r[4]-r[3])}  total {int(r[4]-r[0])} cycles; start(realtime 100MHz) {int(r[5])}")
    names = ["dma issue", "softmax", "barrier(Y end)", "X: PV+QK", "vmcnt(0)", "barrier(X end)", "loop back"]
    for w in range(8):
        rows = st[w, 2:min(nt - 2, 31)]
        if rows.numel() == 0 or (rows[:, 0] == 0).all():
            continue
        seg = [(rows[:, i + 1] - rows[:, i]).float().mean().item() for i in range(6)]
        tile = (rows[1:, 0] - rows[:-1, 0]).float().mean().item() if rows.shape[0] > 1 else float("nan")
        print(f"  wave {w}: " + "  ".join(f"{n} {v:7.0f}" for n, v in zip(names, seg)) + f"  | tile {tile:7.0f}")

run("sam global", 32, 16, 1024, 96, 80, False, (32, 32))
run("llama", 4, 32, 703, 128, 128, True, None)
run("clip", 32, 16, 577, 64, 64, False, None)
run("hs128 L2048", 4, 32, 2048, 128, 128, False, None)
