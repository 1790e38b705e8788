import sys, torch
sys.path.insert(0, "/root/repo")
from grove_amd import _lib, ops
L = _lib.lib(); dev = torch.device("cuda:0"); bf16 = torch.bfloat16
M, I, K = 1400, 11008, 4096
g = torch.Generator().manual_seed(M + I)
dy = (torch.randn(M, K, generator=g) * 0.5).to(bf16).to(dev)
w = (torch.randn(I, K, generator=g) * 0.05).to(bf16).to(dev)
gu = (torch.randn(M, 2 * I, generator=g) * 1.5).to(bf16).to(dev)
for mode in (1, 0):
    L.grove_gemm_set_stream_k(mode)
    da = ops.linear(dy, w); v0, s0 = L.grove_gemm_last_variant(), L.grove_gemm_last_stream_k()
    want = ops.swiglu_bwd(gu, da, I)
    out = ops.linear(dy, w, act=ops.ACT_SWIGLU_BWD, residual=gu); v1, s1 = L.grove_gemm_last_variant(), L.grove_gemm_last_stream_k()
    bad = (out != want).nonzero()
    print("mode", mode, "plain variant", v0, s0, "fused", v1, s1, "mismatches", bad.shape[0])
    if bad.shape[0]:
        r, c = bad[:, 0], bad[:, 1]
        print(" rows", r.min().item(), r.max().item(), "cols", c.min().item(), c.max().item(), "first", bad[:5].tolist())
        i, j = bad[0].tolist()
        jj = j % I
        print(" da", da[i, jj].item(), "g", gu[i, jj].item(), "u", gu[i, I + jj].item(), "out", out[i, j].item(), "want", want[i, j].item())
