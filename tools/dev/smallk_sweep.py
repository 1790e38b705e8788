import sys, torch
sys.path.insert(0, "/root/repo")
from grove_amd import ops, _lib
dev = torch.device("cuda:0"); bf = torch.bfloat16
L = _lib.lib()
bad = 0
for tile_m in (192, 256):
    L.grove_gemm_set_tile_m(tile_m)
    for K in (64, 128, 192, 256, 320, 448, 576):
        for (M, N) in ((2812, 4096), (600, 520), (8200, 1280)):
            g = torch.Generator().manual_seed(K + M)
            x = torch.randn(M, K, generator=g).to(bf); w = (torch.randn(N, K, generator=g) * 0.05).to(bf)
            y = ops.linear(x.to(dev), w.to(dev))
            ref = x.float() @ w.float().t()
            err = (y.float().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            v = L.grove_gemm_last_variant() if hasattr(L, "grove_gemm_last_variant") else -1
            if err > 8e-3:
                bad += 1
                print("BAD", tile_m, M, N, K, err, v)
    # SwiGLU pair + bwd at small K
    for K in (64, 128, 320):
        M, I = 1000, 512
        g = torch.Generator().manual_seed(K)
        h = torch.randn(M, K, generator=g).to(bf); wgu = (torch.randn(2 * I, K, generator=g) * 0.05).to(bf)
        a = ops.linear(h.to(dev), ops.swiglu_interleave(wgu.to(dev)), act=ops.ACT_SWIGLU_PAIR)
        gu = h.float() @ wgu.float().t()
        gt, up = gu[:, :I].to(bf).float(), gu[:, I:].to(bf).float()
        ref = torch.nn.functional.silu(gt) * up
        err = (a.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
        if err > 1e-2:
            bad += 1
            print("BAD swiglu", tile_m, K, err)
L.grove_gemm_set_tile_m(0)
print("small-K sweep: bad =", bad)
