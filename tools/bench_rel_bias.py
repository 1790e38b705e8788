"""rel-pos streams (grove_rel_bias_*) vs the GEMM batched over query positions, SAM-H window and global shapes."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
from grove_amd.model.sam import _rcat_tables
dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def timeit(f, n=20):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for name, nb, size in (("window 14x14", 288, 14), ("global 32x32", 32, 32)):
    nh, hd, hp = 16, 80, 96
    L = size * size
    rel_h = (torch.randn(2 * size - 1, hd) * 0.5).to(bf16).to(dev)
    rel_w = (torch.randn(2 * size - 1, hd) * 0.5).to(bf16).to(dev)
    rcat, rcat_t, khp, rel_ld = _rcat_tables(size, rel_h, rel_w, hd, hp, hd ** -0.5)
    ld = 3 * nh * hp
    qkv = (torch.randn(nb * L, ld, device=dev) * 0.5).to(bf16)
    dq = torch.zeros_like(qkv)
    hrow = (torch.arange(nb, dtype=torch.int32)[:, None] * (L * (ld // hp)) + torch.arange(nh, dtype=torch.int32)[None, :]).reshape(-1).to(dev)
    rel = torch.empty((nb * nh, L, rel_ld), dtype=bf16, device=dev)
    t_new = timeit(lambda: ops.rel_bias_fwd(qkv, rcat, nb, nh, L, hp, hd, out=rel))
    t_old = timeit(lambda: ops.gemm_raw(qkv, rcat, rel, nb * nh, rel_ld, hp, hp, hp, L * rel_ld, a_idx=hrow, batch=(L, 1), sA=(ld, 0), sB=(rel_ld * hp, 0), sC=(rel_ld, 0)))
    byts = nb * L * nh * (hp + rel_ld) * 2
    print(f"{name} fwd: stream {t_new:6.1f} us ({byts / t_new / 1e6:.2f} TB/s)   batched GEMM {t_old:6.1f} us")
    drel = (torch.randn(nb * nh, L, rel_ld, device=dev) * 0.3).to(bf16)
    t_new = timeit(lambda: ops.rel_bias_bwd(drel, rcat_t, dq, nb, nh, L, hp, hd))
    t_old = timeit(lambda: ops.gemm_raw(drel, rcat_t, dq, nb * nh, hp, rel_ld, L * rel_ld, rel_ld, hp, c_idx=hrow, residual=dq, ldr=hp,
                                        batch=(L, 1), sA=(rel_ld, 0), sB=(hp * rel_ld, 0), sC=(ld, 0), sR=(ld, 0)))
    byts = nb * L * nh * (2 * hd + rel_ld) * 2
    print(f"{name} bwd: stream {t_new:6.1f} us ({byts / t_new / 1e6:.2f} TB/s)   batched GEMM {t_old:6.1f} us")
