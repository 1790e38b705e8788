"""Flash-attention micro-benchmark on the four attention shapes of the GROVE step (random data)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
cases = [("sam window", 288, 16, 196, 96, 80, False, (14, 14)), ("sam win gen", 288, 16, 196, 96, 80, False, (14, 14)), ("sam global", 32, 16, 1024, 96, 80, False, (32, 32)),
         ("llama", 4, 32, 703, 128, 128, True, None), ("clip", 32, 16, 577, 64, 64, False, None)]
for name, B, H, L, hs, hd, causal, rel_hw in cases:
    qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
    qkv = qkv.to(bf)
    do = torch.randn(B * L, H * hs, device=dev).to(bf)
    rel, arg = None, (0, 0)
    if rel_hw:
        khp = (rel_hw[0] + 15) // 16 * 16
        rel = torch.randn(B * H, L, 2 * khp, device=dev).to(bf)
        arg = (khp, rel_hw[1])
    alpha = hd ** -0.5
    dq = torch.empty_like(qkv)
    hv = hd if hd < hs else 0
    from grove_amd import _lib
    _lib.lib().grove_flash_attn_set_window_kernels(0 if name == "sam win gen" else 1)  # "gen": the general kernels on the window shape
    def fwd():
        return ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=causal, rel=rel, rel_hw=arg, want_lse=True, hs_valid=hv)
    out, lse = fwd()
    def bwd():
        ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=causal, rel=rel, rel_hw=arg, want_drel=rel is not None, hs_valid=hv)
    res = []
    for fn, mult in ((fwd, 4), (bwd, 10)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flops = mult * B * H * L * L * hd * (0.5 if causal else 1.0)
        res.append((ms, flops / ms / 1e9))
    print(f"{name:11s} B={B} H={H} L={L} hs={hs}: fwd {res[0][0]:.3f} ms {res[0][1]:6.1f} TF/s | bwd {res[1][0]:.3f} ms {res[1][1]:6.1f} TF/s", flush=True)

# SAM window attention as the step runs it: real tokens only (q_valid, pad_row) and o / d_o in token order with compact heads (o_map)
_lib.lib().grove_flash_attn_set_window_kernels(1)
F, H, L, hs, hd, ws, grid = 32, 16, 196, 96, 80, 14, 32
nw = (grid + ws - 1) // ws
B = F * nw * nw
ext = [min(ws, grid - w * ws) for w in range(nw)]
qv = torch.tensor([[ext[wy], ext[wx]] for _ in range(F) for wy in range(nw) for wx in range(nw)], dtype=torch.int32, device=dev)
omap = torch.full((B, ws, ws), -1, dtype=torch.int32)
for f in range(F):
    for wy in range(nw):
        for wx in range(nw):
            ys = torch.arange(ext[wy]) + wy * ws
            xs = torch.arange(ext[wx]) + wx * ws
            omap[(f * nw + wy) * nw + wx, :ext[wy], :ext[wx]] = (f * grid * grid + ys[:, None] * grid + xs[None, :]).to(torch.int32)
omap = omap.reshape(-1).to(dev)
ntok = F * grid * grid
qkv = torch.zeros(B * L, 3 * H * hs, device=dev)
qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, device=dev)
qkv = qkv.to(bf)
pad_row = qkv[0].clone()
rel = torch.randn(B * H, L, 32, device=dev).to(bf)
dq = torch.empty_like(qkv)
alpha = hd ** -0.5
for name, tokens in (("win real", False), ("win real tok", True)):
    kw = dict(o_map=omap, o_rows=ntok) if tokens else {}
    do = torch.randn(ntok, H * hd, device=dev).to(bf) if tokens else torch.randn(B * L, H * hs, device=dev).to(bf)
    def fwd():
        return ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel, rel_hw=(16, ws), want_lse=True, hs_valid=hd, q_valid=qv, pad_row=pad_row, **kw)
    out, lse = fwd()
    def bwd():
        ops.flash_attn_bwd(qkv, out, do, lse, dq, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel, rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv,
                           pad_row=pad_row, **({"o_map": omap} if tokens else {}))
    res = []
    for fn in (fwd, bwd):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    print(f"{name:12s} F={F} windows={B} H={H} (q_valid, pad_row{', o_map' if tokens else ''}): fwd {res[0]:.3f} ms | bwd {res[1]:.3f} ms", flush=True)

# round 6b: the rel-pos terms made inside the window kernels (grove_flash_attn_params.rel_table) against the two streams they replace
from grove_amd.model.sam import _rcat_tables
rel_h = (torch.randn(2 * ws - 1, hd, device=dev) * 0.5).to(bf)
rel_w = (torch.randn(2 * ws - 1, hd, device=dev) * 0.5).to(bf)
T = ops.rel_table_images(rel_h, rel_w, ws, alpha)
rcat, rcat_t, _, _ = _rcat_tables(ws, rel_h, rel_w, hd, hs, alpha)
do = torch.randn(ntok, H * hd, device=dev).to(bf)
dg = torch.empty((ntok, 3 * H * hd), dtype=bf, device=dev)
rel_o = torch.empty((B * H, L, 32), dtype=bf, device=dev)
out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel_hw=(16, ws), want_lse=True, hs_valid=hd, q_valid=qv, pad_row=pad_row,
                          o_map=omap, o_rows=ntok, rel_table=T, rel_out=rel_o)
cases = {
    "rel_bias_fwd stream": lambda: ops.rel_bias_fwd(qkv, rcat, B, H, L, hs, hd, out=rel_o, q_valid=qv, kw=ws),
    "rel_bias_bwd stream": lambda: ops.rel_bias_bwd(rel_o, rcat_t, dg, B, H, L, hs, hd, q_valid=qv, kw=ws, dq_map=omap),
    "fwd two-stream form (tok)": lambda: ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel, rel_hw=(16, ws), want_lse=True, hs_valid=hd,
                                                        q_valid=qv, pad_row=pad_row, o_map=omap, o_rows=ntok, out=out),
    "fwd rel in kernel, operand kept": lambda: ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel_hw=(16, ws), want_lse=True, hs_valid=hd, q_valid=qv,
                                                              pad_row=pad_row, o_map=omap, o_rows=ntok, rel_table=T, rel_out=rel_o, out=out),
    "fwd rel in kernel, inference": lambda: ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel_hw=(16, ws), hs_valid=hd, q_valid=qv,
                                                           pad_row=pad_row, o_map=omap, o_rows=ntok, rel_table=T, out=out),
    "bwd two-stream form (g_tok)": lambda: ops.flash_attn_bwd(qkv, out, do, lse, dg, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel, rel_hw=(16, ws), want_drel=True,
                                                              hs_valid=hd, q_valid=qv, pad_row=pad_row, o_map=omap, grads_tok=True),
    "bwd rel in kernel (g_tok)": lambda: ops.flash_attn_bwd(qkv, out, do, lse, dg, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel_o, rel_hw=(16, ws), hs_valid=hd,
                                                            q_valid=qv, pad_row=pad_row, o_map=omap, grads_tok=True, rel_table=T),
}
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"  {name:34s} {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us", flush=True)
