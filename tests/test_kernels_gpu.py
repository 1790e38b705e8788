"""Per-kernel numerics of libgrove_hip.so against plain PyTorch fp32 references on the CPU.

Every test calls through the C-ABI (grove_amd.ops -> ctypes -> include/grove_hip.h). Inputs are
rounded to bf16 first, so the fp32 reference sees exactly the values the kernel sees; the
tolerance is the bf16 output rounding (2^-8 relative) unless the output is fp32.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf16)


def close(out, ref, rtol, what=""):
    out = out.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    err = (out - ref).abs().max().item()
    lim = rtol * max(ref.abs().max().item(), 1e-6)
    assert err <= lim, f"{what}: max abs err {err:.4g} > {lim:.4g}"


def act_ref(x, act):
    from grove_amd import ops
    if act == ops.ACT_RELU:
        return F.relu(x)
    if act == ops.ACT_GELU:
        return F.gelu(x)
    if act == ops.ACT_QUICKGELU:
        return x * torch.sigmoid(1.702 * x)
    if act == ops.ACT_SILU:
        return F.silu(x)
    return x


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("staging", [1, 0])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (577, 1024, 1024), (100, 72, 96), (1, 4, 256),
                                   (300, 1000, 608), (130, 257, 32)])
def test_gemm_plain(dev, M, N, K, staging):
    from grove_amd import ops
    ops.gemm_set_staging(bool(staging))
    try:
        a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
        ref = a.float() @ b.float().t()
        out32 = ops.linear(a.to(dev), b.to(dev), out_dtype=torch.float32)
        close(out32, ref, 2e-5, "f32 out")
        out16 = ops.linear(a.to(dev), b.to(dev))
        close(out16, ref, 6e-3, "bf16 out")
    finally:
        ops.gemm_set_staging(True)


def test_gemm_asymmetric_identity(dev):
    # A = I with an asymmetric B catches a transposed C write (cdna guide §3)
    from grove_amd import ops
    K = 128
    a = torch.eye(K).to(bf16)
    b = (torch.arange(K * K).reshape(K, K) % 251).float().to(bf16)
    out = ops.linear(a.to(dev), b.to(dev), out_dtype=torch.float32)
    close(out, b.float().t(), 0.0 + 1e-7, "A=I")


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm_epilogue(dev, act):
    from grove_amd import ops
    M, N, K = 200, 320, 192
    a, b = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.1)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    alpha_t = torch.tensor([0.37])
    pre = a.float() @ b.float().t() + bias.float()
    ref = act_ref(pre, act) * math.tanh(0.37) + res.float()
    aux = torch.empty(M, N, dtype=bf16, device=dev)
    out = ops.linear(a.to(dev), b.to(dev), bias.to(dev), act=act, residual=res.to(dev), aux=aux,
                     scale_ptr=alpha_t.to(dev), scale_tanh=True)
    close(out, ref, 8e-3, "epilogue out")
    close(aux, pre, 6e-3, "aux pre-activation")


@pytest.mark.parametrize("M,N,K", [(200, 320, 192), (4096, 1280, 1024)])  # the simple and the pipelined kernel
@pytest.mark.parametrize("act", [2, 3])
def test_gemm_fused_activation_backward(dev, act, M, N, K):
    """MLP backward without the elementwise pass: the forward GEMM stores act'(pre-activation) as aux (aux_grad), the
    dgrad GEMM of the next layer multiplies by it (residual_mul) — together dX = (dY @ W2) * act'(x) as autograd computes it."""
    from grove_amd import ops
    a, b = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.05)
    bias = rnd(N, seed=5)
    pre = (a.float() @ b.float().t() + bias.float()).requires_grad_(True)
    y = act_ref(pre, act)
    dy = rnd(M, N, seed=7)
    y.backward(dy.float())
    grad = pre.grad / dy.float()  # act'(pre)
    grad = torch.where(dy.float() == 0, torch.zeros_like(grad), grad)
    aux = torch.empty(M, N, dtype=bf16, device=dev)
    out = ops.linear(a.to(dev), b.to(dev), bias.to(dev), act=act, aux=aux, aux_grad=True)
    close(out, y, 8e-3, "activation out")
    mask = dy.float() != 0
    assert ((aux.float().cpu() - grad).abs()[mask].max().item()) <= 8e-3 * 1.2, "aux = act'(pre)"
    # second half: C = (dZ @ W) * aux
    dz, w2 = rnd(M, K, seed=8), rnd(N, K, seed=9, scale=0.05)
    ref = (dz.float() @ w2.float().t()) * aux.float().cpu()
    got = ops.linear(dz.to(dev), w2.to(dev), residual=aux, residual_mul=True)
    close(got, ref, 8e-3, "dgrad * act'")


@pytest.mark.parametrize("tile_m", [256, 193])
def test_pipelined_tile_map_covers_every_tile(dev, tile_m):
    """The persistent kernel maps tile index -> origin with multiply-high reciprocals (bands of 8 tile rows, a shorter last
    band): every count of tile rows modulo 8 — including a last band of ONE row, whose reciprocal 2^32 does not fit in 32 bits —
    must write every output tile exactly as the simple kernel does. The output starts as NaN so a skipped tile shows."""
    from grove_amd import _lib, ops
    L = _lib.lib()
    bm = 256 if tile_m == 256 else 192
    K, N = 128, 776  # 4 column tiles, the last one partial
    b = rnd(N, K, seed=12, scale=0.1).to(dev)
    try:
        for tiles_m in list(range(1, 19)) + [33, 73]:
            M = bm * (tiles_m - 1) + 8
            a = rnd(M, K, seed=11).to(dev)
            L.grove_gemm_set_tile_m(128)
            ref = ops.linear(a, b)
            L.grove_gemm_set_tile_m(tile_m)
            out = torch.full((M, N), float("nan"), dtype=bf16, device=dev)
            ops.linear(a, b, out=out)
            assert L.grove_gemm_last_variant() in (4, 5), "the pipelined kernel must have run"
            assert torch.equal(out, ref), f"{tiles_m} tile rows (mod 8 = {tiles_m % 8})"
    finally:
        L.grove_gemm_set_tile_m(0)


STEP_SHAPES = [(32768, 5120, 1280), (32768, 1280, 5120), (32768, 3840, 1280), (32768, 1280, 1280), (18464, 4096, 1024),
               (18464, 1024, 4096), (18464, 3072, 1024), (18464, 1024, 1024), (2812, 22016, 4096), (2812, 4096, 11008),
               (2812, 12288, 4096), (2812, 4096, 4096), (2304, 4096, 1024)]


@pytest.mark.parametrize("M,N,K", STEP_SHAPES)
def test_step_shapes_dispatch_matches_simple_kernel(dev, M, N, K):
    """The GEMM shapes of the full-size training step (profiles/*_gemm_shapes.txt), as the cost model dispatches them, bit for bit
    against the simple two-barrier kernel — the sizes at which tile-count-dependent paths (tile map bands, persistent rounds,
    edge tiles) actually differ from the small test shapes. Bias + residual epilogue; the output starts as NaN."""
    from grove_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(bf16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(bf16).to(dev)
    bias = torch.randn(N, generator=g).to(bf16).to(dev)
    res = torch.randn(M, N, generator=g).to(bf16).to(dev)
    try:
        L.grove_gemm_set_tile_m(128)
        ref = ops.linear(a, b, bias, residual=res)
        L.grove_gemm_set_tile_m(0)
        out = torch.full((M, N), float("nan"), dtype=bf16, device=dev)
        ops.linear(a, b, bias, residual=res, out=out)
        assert L.grove_gemm_last_variant() in (4, 5), "these shapes belong to the pipelined kernels"
        if L.grove_gemm_last_stream_k():  # a split K sum rounds differently: bit-exact with whole tiles, within bf16 rounding as dispatched
            close(out, ref, 2 ** -7, "stream-K dispatch")
            L.grove_gemm_set_stream_k(0)
            out = torch.full((M, N), float("nan"), dtype=bf16, device=dev)
            ops.linear(a, b, bias, residual=res, out=out)
        assert torch.equal(out, ref)
    finally:
        L.grove_gemm_set_tile_m(0)
        L.grove_gemm_set_stream_k(1)


@pytest.mark.parametrize("tile_m,M", [(256, 20200), (193, 15350)])
def test_gemm_stream_k_tail(dev, tile_m, M):
    """Stream-K tail of the pipelined kernels (one whole round + 60 / 64 tiles cut into four K ranges): every epilogue family
    through the fix-up launch — plain bias + residual, GELU with aux and a tanh'd scale, the SwiGLU pair, scattered fp32 rows
    (c_idx / r_idx), padded-head output columns (n_map) — against the same launch with whole tiles (bf16 rounding of a
    different fp32 sum order) and against fp32. The outputs start as NaN: a tile nobody finished shows."""
    from grove_amd import _lib, ops
    L = _lib.lib()
    N, K = 1000, 2560
    g = torch.Generator().manual_seed(M)
    a = (torch.randn(M, K, generator=g) * 0.5).to(bf16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(bf16).to(dev)
    bias = torch.randn(N, generator=g).to(bf16).to(dev)
    res = torch.randn(M, N, generator=g).to(bf16).to(dev)
    perm = torch.randperm(M, generator=g).to(torch.int32).to(dev)
    sc = torch.tensor([0.7]).to(dev)
    rows = slice(M - 1500, M)  # the last rows: tiles of the split round

    def run(**kw):
        outs = []
        for mode in (0, 1):
            L.grove_gemm_set_stream_k(mode)
            kw2 = dict(kw)
            if "aux" in kw2:
                kw2["aux"] = torch.full_like(kw2["aux"], float("nan"))
            if "out" in kw2:
                kw2["out"] = torch.full_like(kw2["out"], float("nan"))
            w = kw2.pop("w", b)
            o = ops.linear(a, w, kw2.pop("bias", bias), **kw2)
            assert L.grove_gemm_last_variant() == (4 if tile_m == 256 else 5)
            assert (L.grove_gemm_last_stream_k() > 0) == (mode == 1), "the split must engage exactly when it is on"
            outs.append((o, kw2.get("aux")))
        return outs

    try:
        L.grove_gemm_set_tile_m(tile_m)
        pre = a[rows].float() @ b.float().t() + bias.float()
        # plain
        (o0, _), (o1, _) = run(residual=res)
        close(o1, o0, 2 ** -7, "plain: split vs whole")
        close(o1[rows], pre + res[rows].float(), 2 ** -7, "plain vs fp32")
        # GELU, aux = pre-activation, tanh'd scale
        (o0, x0), (o1, x1) = run(act=ops.ACT_GELU, aux=torch.empty(M, N, dtype=bf16, device=dev), scale_ptr=sc, scale_tanh=True)
        close(o1, o0, 2 ** -7, "gelu: split vs whole")
        close(x1, x0, 2 ** -7, "gelu aux: split vs whole")
        close(o1[rows], torch.nn.functional.gelu(pre) * torch.tanh(sc.float()), 2 ** -6, "gelu vs fp32")
        # SwiGLU pair: gate / up rows interleaved in groups of four
        Np = 1024
        wg, wu = (torch.randn(Np // 2, K, generator=g) * 0.05).to(bf16), (torch.randn(Np // 2, K, generator=g) * 0.05).to(bf16)
        wp = torch.stack([wg.view(-1, 4, K), wu.view(-1, 4, K)], 1).reshape(Np, K).to(dev)
        (o0, _), (o1, _) = run(w=wp, bias=None, act=ops.ACT_SWIGLU_PAIR)
        close(o1, o0, 2 ** -7, "swiglu pair: split vs whole")
        gate, up = a[rows].float() @ wg.float().t().to(dev), a[rows].float() @ wu.float().t().to(dev)
        close(o1[rows], torch.nn.functional.silu(gate) * up, 2 ** -6, "swiglu pair vs fp32")
        # scattered fp32 rows: the wide epilogue
        (o0, _), (o1, _) = run(out=torch.empty(M, N, dtype=torch.float32, device=dev), c_idx=perm, r_idx=perm, residual=res)
        close(o1, o0, 1e-5, "scattered fp32: split vs whole")
        full = torch.zeros(M, N, device=dev)
        full[perm[rows].long()] = pre + res[perm[rows].long()].float()
        close(o1[perm[rows].long()], full[perm[rows].long()], 1e-4, "scattered fp32 vs fp32")
        # padded-head output columns: groups of 40 columns padded by 8
        (o0, _), (o1, _) = run(n_map=(40, 8), out=torch.empty(M, N // 40 * 48, dtype=bf16, device=dev))
        close(o1, o0, 2 ** -7, "n_map: split vs whole")
        close(o1.view(M, -1, 48)[rows, :, :40].reshape(-1, N), pre, 2 ** -7, "n_map vs fp32")
        assert (o1.view(M, -1, 48)[:, :, 40:] == 0).all(), "n_map pad columns"
    finally:
        L.grove_gemm_set_tile_m(0)
        L.grove_gemm_set_stream_k(1)


@pytest.mark.parametrize("M,N,K", [(260, 4096, 22016), (260, 4096, 4096), (12, 4096, 4096), (615, 4096, 4096), (576, 256, 2048), (260, 4096, 1024)])
def test_gemm_few_tiles_long_k_is_cut_over_the_idle_cus(dev, M, N, K):
    """Round 4: fewer output tiles than CUs with a long K (the few-row GEMMs of LLaMA's last-layer tail, the 615-row prefill, the
    decoder's K = 2048 layers): the AUTOMATIC dispatch takes the persistent kernel and cuts the only round into up to 8 K ranges
    (stream-K parts + the fix-up launch) instead of the two-barrier kernel on a fraction of the chip; K = 1024 stays whole.
    Against fp32 and against the launch with the cut off; outputs start as NaN."""
    from grove_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(bf16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(bf16).to(dev)
    bias = torch.randn(N, generator=g).to(bf16).to(dev)
    res = torch.randn(M, N, generator=g).to(bf16).to(dev)
    outs = []
    try:
        for mode in (1, 0):
            L.grove_gemm_set_stream_k(mode)
            o = ops.linear(a, b, bias, residual=res, out=torch.full((M, N), float("nan"), dtype=bf16, device=dev))
            outs.append((o, L.grove_gemm_last_variant(), L.grove_gemm_last_stream_k()))
    finally:
        L.grove_gemm_set_stream_k(1)
    (o1, v1, s1), (o0, v0, s0) = outs
    if K >= 2048:
        assert v1 in (4, 5) and s1 >= 4 and (v0 not in (4, 5) or s0 == 0), (v1, s1, v0, s0)  # (last_stream_k is the last PIPELINED launch's)
    else:
        assert v1 not in (4, 5) or s1 == 0
    want = a.float() @ b.float().t() + bias.float() + res.float()
    close(o1, want, 2 ** -7, "few tiles, cut K vs fp32")
    close(o1, o0, 2 ** -7, "few tiles, cut K vs whole tiles")


@pytest.mark.parametrize("M,I,K", [(2812, 11008, 4096), (1400, 11008, 4096), (260, 2752, 1024), (703, 1024, 512)])
def test_swiglu_backward_in_the_dgrad_epilogue(dev, M, I, K):
    """Round 4: GROVE_ACT_SWIGLU_BWD — d(gate | up) of LlamaMLP's silu(gate) * up computed in the epilogue of the down-projection's dgrad
    GEMM (C [M, 2I] from d a = dy . W^T [M, I] and the saved gate | up [M, 2I]). BIT-identical to the two launches it replaces (the GEMM to
    bf16, then grove_swiglu_bwd) — whole tiles, edge rows, the stream-K fix-up path ((1400, 11008): one round + 2 tiles) and the few-tile
    cut — and against torch autograd of the product in fp32."""
    from grove_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + I)
    dy = (torch.randn(M, K, generator=g) * 0.5).to(bf16).to(dev)
    w = (torch.randn(I, K, generator=g) * 0.05).to(bf16).to(dev)           # W_down^T: [I, H]
    gu = (torch.randn(M, 2 * I, generator=g) * 1.5).to(bf16).to(dev)
    try:  # both forms on the SAME kernel and plan (left to itself the plain GEMM of a shape may take another kernel: another fp32 sum order)
        for tile in (193, 256):
            L.grove_gemm_set_tile_m(tile)
            da = ops.linear(dy, w)
            plan = (L.grove_gemm_last_variant(), L.grove_gemm_last_stream_k())
            want = ops.swiglu_bwd(gu, da, I)
            out = ops.linear(dy, w, act=ops.ACT_SWIGLU_BWD, residual=gu, out=torch.full((M, 2 * I), float("nan"), dtype=bf16, device=dev))
            assert (L.grove_gemm_last_variant(), L.grove_gemm_last_stream_k()) == plan and plan[0] in (4, 5)
            assert torch.equal(out, want), (tile, (out.float() - want.float()).abs().max().item())
    finally:
        L.grove_gemm_set_tile_m(0)
    out = ops.linear(dy, w, act=ops.ACT_SWIGLU_BWD, residual=gu, out=torch.full((M, 2 * I), float("nan"), dtype=bf16, device=dev))  # automatic dispatch
    assert L.grove_gemm_last_variant() in (4, 5)
    close(out, want, 2 ** -6, "automatic dispatch vs the two launches")
    gate, up = gu[:, :I].float().requires_grad_(), gu[:, I:].float().requires_grad_()
    (torch.nn.functional.silu(gate) * up).backward(da.float())
    close(out[:, :I], gate.grad, 2 ** -7, "d gate vs autograd")
    close(out[:, I:], up.grad, 2 ** -7, "d up vs autograd")
    with pytest.raises(RuntimeError):
        ops.linear(dy, w, act=ops.ACT_SWIGLU_BWD)        # no gate | up operand


def test_gemm_stream_k_gathered_taps(dev):
    """The gathered-A instances under the stream-K tail: a K range that starts inside the tap list (27-tap Conv3d rows, -1 = zero
    row; 2 K tiles per tap, parts of 14 K tiles) with the ReLU + residual + tanh'd scale epilogue the adapters use."""
    from grove_amd import _lib, ops
    from grove_amd.model.indexing import conv3d_gather_index
    L = _lib.lib()
    G, T, H, W, Ci, Co = 2, 8, 32, 40, 128, 1024
    M = G * T * H * W  # 20480 rows: 80 x 4 tiles of 256 = one round + 64
    x, w, bias = rnd(M, Ci, seed=21).to(dev), rnd(Co, 27 * Ci, seed=22, scale=0.03).to(dev), rnd(Co, seed=23).to(dev)
    res = rnd(M, Co, seed=24).to(dev)
    idx = conv3d_gather_index(G, T, H, W).to(dev)
    kw = dict(act=ops.ACT_RELU, residual=res, scale_ptr=torch.tensor([0.3]).to(dev), scale_tanh=True, a_idx=idx, a_taps=27, M=M)
    try:
        for tile_m, variant in ((256, 6), (193, 7)):
            L.grove_gemm_set_tile_m(tile_m)
            outs = []
            for mode in (0, 2):  # (2: the 192-row tiling leaves 108 x 4 = 432 tiles = 1 round + 176: no split there)
                L.grove_gemm_set_stream_k(mode)
                outs.append(ops.linear(x, w, bias, **kw))
                assert L.grove_gemm_last_variant() == variant
                if tile_m == 256:
                    assert (L.grove_gemm_last_stream_k() > 0) == (mode == 2)
            close(outs[1], outs[0], 2 ** -7, f"gathered taps, tile {tile_m}: split vs whole")
        L.grove_gemm_set_tile_m(128)
        L.grove_gemm_set_stream_k(1)
        ref = ops.linear(x, w, bias, **kw)
        close(outs[1], ref, 2 ** -7, "gathered taps vs the two-barrier kernel")
    finally:
        L.grove_gemm_set_tile_m(0)
        L.grove_gemm_set_stream_k(1)


def test_gemm_accumulate_and_alpha(dev):
    from grove_amd import ops
    M, N, K = 96, 160, 64
    a, b = rnd(M, K, seed=7), rnd(N, K, seed=8)
    c0 = torch.randn(M, N, generator=torch.Generator().manual_seed(9))
    out = c0.clone().to(dev)
    ops.linear(a.to(dev), b.to(dev), out=out, accumulate=True, alpha=0.5)
    close(out, c0 + 0.5 * (a.float() @ b.float().t()), 2e-5, "accumulate")


def test_gemm_batched_strided(dev):
    # attention-shaped: q/k packed as [B, S, 3, H, d]; scores[b,h] = q k^T
    from grove_amd import ops
    B, S, H, d = 2, 77, 4, 64
    qkv = rnd(B, S, 3 * H * d, seed=10)
    q = qkv[..., :H * d].reshape(B, S, H, d)
    k = qkv[..., H * d:2 * H * d].reshape(B, S, H, d)
    ref = torch.einsum("bshd,bthd->bhst", q.float(), k.float()) * 0.125
    ld_s = ops.pad_to(S, 4)
    x = qkv.to(dev)
    scores = torch.empty(B * H, S, ld_s, dtype=torch.float32, device=dev)
    ops.gemm_raw(x, x[:, :, H * d:], scores, S, S, d, 3 * H * d, 3 * H * d, ld_s, batch=(B, H),
                 sA=(S * 3 * H * d, d), sB=(S * 3 * H * d, d), sC=(H * S * ld_s, S * ld_s), alpha=0.125)
    close(scores[:, :, :S].reshape(B, H, S, S), ref, 2e-5, "batched scores")


def test_gemm_gather_conv3d(dev):
    # implicit-GEMM Conv3d 3x3x3 'same' over channels-last tokens (SAM / CLIP adapters)
    from grove_amd import ops
    from grove_amd.model.indexing import conv3d_gather_index
    G, T, H, W, Ci, Co = 1, 3, 4, 5, 64, 48
    x = rnd(G * T * H * W, Ci, seed=11)
    w = rnd(Co, Ci, 3, 3, 3, seed=12, scale=0.05)
    bias = rnd(Co, seed=13)
    xr = x.float().reshape(G, T, H, W, Ci).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xr, w.float(), bias.float(), padding=1).permute(0, 2, 3, 4, 1).reshape(-1, Co)
    ref = torch.relu(ref) * math.tanh(0.3) + x.float()[:, :Co]
    idx = conv3d_gather_index(G, T, H, W).to(dev)          # [27, M]
    wp = w.permute(0, 2, 3, 4, 1).reshape(Co, 27 * Ci).contiguous()  # [Co, tap, Ci]
    res = x[:, :Co].contiguous()
    out = ops.linear(x.to(dev), wp.to(dev), bias.to(dev), act=ops.ACT_RELU, residual=res.to(dev),
                     scale_ptr=torch.tensor([0.3]).to(dev), scale_tanh=True, a_idx=idx, a_taps=27, M=x.shape[0])
    close(out, ref, 8e-3, "conv3d implicit gemm")


def test_gemm_scatter_rows(dev):
    from grove_amd import ops
    M, N, K = 70, 64, 64
    a, b = rnd(M, K, seed=14), rnd(N, K, seed=15)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(1)).int()
    c_idx = perm.clone()
    c_idx[::7] = -1
    r_idx = (torch.arange(M) % 5).int()
    res = rnd(5, N, seed=16)
    out = torch.zeros(M, N, dtype=bf16, device=dev)
    ops.linear(a.to(dev), b.to(dev), out=out, c_idx=c_idx.to(dev), r_idx=r_idx.to(dev), residual=res.to(dev))
    ref = torch.zeros(M, N)
    full = a.float() @ b.float().t() + res.float()[r_idx.long()]
    for m in range(M):
        if c_idx[m] >= 0:
            ref[c_idx[m]] = full[m]
    close(out, ref, 6e-3, "scatter")


def test_transpose(dev):
    from grove_amd import ops
    x = rnd(3, 77, 130, seed=17)
    out = torch.full((3, 130, 96), 7.0, dtype=bf16, device=dev)
    ops.transpose(x.to(dev), 77, 130, 130, out, 96, pad_to_cols=96, batch=(3, 1), s_in=(77 * 130, 0), s_out=(130 * 96, 0))
    ref = torch.zeros(3, 130, 96)
    ref[:, :, :77] = x.float().transpose(1, 2)
    close(out, ref, 1e-7, "transpose")


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("C", [64, 256, 1024, 1280, 4096])
def test_layernorm_fwd_bwd(dev, C):
    from grove_amd import ops
    rows = 37
    x, w, b, dy = rnd(rows, C, seed=18), rnd(C, seed=19), rnd(C, seed=20), rnd(rows, C, seed=21)
    xr = x.float().requires_grad_(True)
    wr, br = w.float().requires_grad_(True), b.float().requires_grad_(True)
    y_ref = F.layer_norm(xr, (C,), wr, br, 1e-5)
    y_ref.backward(dy.float())
    y, mean, rstd = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-5, save_stats=True)
    close(y, y_ref, 8e-3, "ln fwd")
    if C <= 2048:
        dw = torch.zeros(C, dtype=torch.float32, device=dev)
        db = torch.zeros(C, dtype=torch.float32, device=dev)
    else:
        dw = db = None
    dx = ops.layernorm_bwd(x.to(dev), w.to(dev), dy.to(dev), mean, rstd, dweight=dw, dbias=db)
    close(dx, xr.grad, 1e-2, "ln dx")
    if dw is not None:
        close(dw, wr.grad, 1e-4, "ln dw")
        close(db, br.grad, 1e-4, "ln db")


def test_layernorm_scatter_f32(dev):
    from grove_amd import ops
    rows, C = 20, 256
    x, w, b = rnd(rows, C, seed=22), rnd(C, seed=23), rnd(C, seed=24)
    idx = (torch.arange(rows) * 2 % 31).int()
    y, _, _ = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, out_idx=idx.to(dev), out_rows=31, out_dtype=torch.float32)
    ref = torch.zeros(31, C)
    ref[idx.long()] = F.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-6)
    close(y, ref, 1e-5, "ln scatter f32")


@pytest.mark.parametrize("C", [128, 4096])
def test_rmsnorm_fwd_bwd(dev, C):
    from grove_amd import ops
    rows = 19
    x, w, dy = rnd(rows, C, seed=25), rnd(C, seed=26), rnd(rows, C, seed=27)
    xr = x.float().requires_grad_(True)
    y_ref = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()
    y_ref.backward(dy.float())
    y = ops.rmsnorm(x.to(dev), w.to(dev), 1e-5)
    close(y, y_ref, 8e-3, "rms fwd")
    dx = ops.rmsnorm_bwd(x.to(dev), w.to(dev), dy.to(dev), 1e-5)
    close(dx, xr.grad, 1e-2, "rms dx")


@pytest.mark.parametrize("C,rms", [(128, True), (4096, True), (1280, False), (256, False)])
def test_norm_residual_stream_form(dev, C, rms):
    """res (fp32) += x (bf16 branch output) fused into the norm that follows; the stream is never rounded: after 40 adds it equals
    the fp32 sum exactly (to fp32 rounding), its bf16 copy is the rounding of that sum, and y is the norm of the fp32 row."""
    from grove_amd import ops
    rows = 23
    w, b = rnd(C, seed=40), rnd(C, seed=41)
    res0 = rnd(rows, C, seed=42).float()
    res = res0.clone().to(dev)
    ref = res0.clone()
    norm = (lambda v: v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()) if rms else \
        (lambda v: F.layer_norm(v, (C,), w.float(), b.float(), 1e-5))
    # first norm of a stack: no pending branch output
    y0 = ops.rmsnorm(None, w.to(dev), 1e-5, res=res) if rms else ops.layernorm(None, w.to(dev), b.to(dev), 1e-5, res=res)[0]
    close(y0, norm(ref), 8e-3, "norm of the untouched stream")
    assert torch.equal(res.cpu(), res0)
    for step in range(40):
        t = (rnd(rows, C, seed=100 + step).float() * 0.05).to(torch.bfloat16)
        ref = ref + t.float()
        xb = torch.empty((rows, C), dtype=torch.bfloat16, device=dev)
        if rms:
            y = ops.rmsnorm(t.to(dev), w.to(dev), 1e-5, res=res, res_bf16=xb)
        else:
            y, mean, rstd = ops.layernorm(t.to(dev), w.to(dev), b.to(dev), 1e-5, res=res, res_bf16=xb, save_stats=True)
    assert (res.cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.equal(xb.cpu(), res.cpu().to(torch.bfloat16))
    close(y, norm(ref), 8e-3, "norm of the fp32 stream")
    if not rms:
        close(mean, ref.mean(-1), 1e-4, "saved mean")
    # stream update without a norm (the consumer needs the stream as a bf16 GEMM operand)
    t = (rnd(rows, C, seed=99).float() * 0.05).to(torch.bfloat16)
    xb2 = torch.empty_like(xb)
    ops.stream_add(res, t.to(dev), res_bf16=xb2)
    ref = ref + t.float()
    assert (res.cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.equal(xb2.cpu(), res.cpu().to(torch.bfloat16))
    ops.stream_add(res, None, res_bf16=xb)  # copy only
    assert torch.equal(xb, xb2)


# ----------------------------------------------------------------------------- softmax family
@pytest.mark.parametrize("Lq,Lk,causal", [(6, 6, False), (577, 577, False), (50, 50, True), (1, 700, True), (196, 196, False)])
def test_softmax_fwd_bwd(dev, Lq, Lk, causal):
    from grove_amd import ops
    batch, heads = 6, 3
    g = torch.Generator().manual_seed(28)
    ld_s = ops.pad_to(Lk, 4)
    s = torch.randn(batch, Lq, ld_s, generator=g) * 3
    kv_len = torch.tensor([Lk, max(1, Lk - 3)], dtype=torch.int32)
    sr = s[:, :, :Lk].clone().requires_grad_(True)
    mask = torch.zeros(batch, Lq, Lk, dtype=torch.bool)
    for b in range(batch):
        mask[b, :, kv_len[b // heads]:] = True
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        mask |= (j > i + (Lk - Lq))[None]
    p_ref = torch.softmax(sr.masked_fill(mask, float("-inf")), -1)
    p = ops.softmax(s.to(dev), Lk, heads=heads, causal=causal, kv_len=kv_len.to(dev))
    close(p[:, :, :Lk], p_ref, 6e-3, "softmax fwd")
    assert (p[:, :, Lk:].float() == 0).all()
    dp = torch.randn(batch, Lq, ld_s, generator=g)
    # backward reference uses the bf16 probabilities the kernel saw
    pb = p[:, :, :Lk].float().cpu()
    ds_ref = pb * (dp[:, :, :Lk] - (pb * dp[:, :, :Lk]).sum(-1, keepdim=True)) * 0.25
    ds = ops.softmax_bwd(dp.to(dev), p, Lk, 0.25)
    close(ds[:, :, :Lk], ds_ref, 8e-3, "softmax bwd")


def test_softmax_relpos(dev):
    from grove_amd import ops
    B, heads, qh, qw, hd = 2, 2, 5, 7, 16
    L = qh * qw
    g = torch.Generator().manual_seed(29)
    q = rnd(B * L, heads * 32, seed=30)  # head stride 32, real hd 16
    Rh = torch.randn(qh, qh, hd, generator=g)
    Rw = torch.randn(qw, qw, hd, generator=g)
    qr = q.float().reshape(B, qh, qw, heads, 32)[..., :hd]
    rel_h = torch.einsum("bhwnc,hkc->bnhwk", qr, Rh).reshape(B * heads, L, qh)
    rel_w = torch.einsum("bhwnc,wkc->bnhwk", qr, Rw).reshape(B * heads, L, qw)
    rel = ops.relpos(q.to(dev), Rh.to(dev), Rw.to(dev), B, heads, (qh, qw), (qh, qw), hd, 32, heads * 32)
    close(rel[:, :, :qh], rel_h, 1e-5, "rel_h")
    close(rel[:, :, qh:], rel_w, 1e-5, "rel_w")
    ld_s = ops.pad_to(L, 4)
    s = torch.randn(B * heads, L, ld_s, generator=g)
    bias = (rel_h[:, :, :, None] + rel_w[:, :, None, :]).reshape(B * heads, L, L)
    p_ref = torch.softmax(s[:, :, :L] + bias, -1)
    p = ops.softmax(s.to(dev), L, heads=heads, rel=rel, rel_hw=(qh, qw))
    close(p[:, :, :L], p_ref, 6e-3, "softmax+rel")
    # backward: drel and dq
    dp = torch.randn(B * heads, L, ld_s, generator=g)
    pb = p[:, :, :L].float().cpu()
    dS = pb * (dp[:, :, :L] - (pb * dp[:, :, :L]).sum(-1, keepdim=True))
    drel_ref = torch.cat([dS.reshape(-1, L, qh, qw).sum(-1), dS.reshape(-1, L, qh, qw).sum(-2)], -1)
    drel = torch.empty(B * heads, L, qh + qw, dtype=torch.float32, device=dev)
    ops.softmax_bwd(dp.to(dev), p, L, 1.0, drel=drel, rel_hw=(qh, qw))
    close(drel, drel_ref, 1e-4, "drel")
    dq0 = rnd(B * L, heads * 32, seed=31)
    dq = dq0.clone().to(dev)
    ops.relpos(q.to(dev), Rh.to(dev), Rw.to(dev), B, heads, (qh, qw), (qh, qw), hd, 32, heads * 32, rel=drel, dq=dq, backward=True)
    d = drel_ref.reshape(B, heads, qh, qw, qh + qw)
    dq_ref = torch.einsum("bnhwk,hkc->bhwnc", d[..., :qh], Rh) + torch.einsum("bnhwk,wkc->bhwnc", d[..., qh:], Rw)
    full = dq0.float().reshape(B, qh, qw, heads, 32).clone()
    full[..., :hd] += dq_ref
    close(dq, full.reshape(B * L, heads * 32), 1e-2, "relpos dq")


def test_rope(dev):
    from grove_amd import ops
    rows, nh, hd = 33, 3, 64
    x = rnd(rows, 3 * nh * hd, seed=32)
    pos = (torch.arange(rows) * 3 % 29).int()
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = pos.float()[:, None] * inv[None]
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1), torch.cat([ang.sin(), ang.sin()], -1)
    ref = x.float().clone()
    for col0 in (0, nh * hd):
        t = x.float()[:, col0:col0 + nh * hd].reshape(rows, nh, hd)
        rot = torch.cat([-t[..., hd // 2:], t[..., :hd // 2]], -1)
        ref[:, col0:col0 + nh * hd] = (t * cos[:, None] + rot * sin[:, None]).reshape(rows, nh * hd)
    y = x.clone().to(dev)
    ops.rope_(y, pos.to(dev), 0, 2 * nh, hd, 10000.0)
    close(y, ref, 6e-3, "rope")
    ops.rope_(y, pos.to(dev), 0, 2 * nh, hd, 10000.0, inverse=True)
    close(y, x.float(), 1.2e-2, "rope inverse")


# ----------------------------------------------------------------------------- elementwise
def test_swiglu_actbwd_add(dev):
    from grove_amd import ops
    rows, I = 21, 88
    gu, dy = rnd(rows, 2 * I, seed=33), rnd(rows, I, seed=34)
    gr = gu.float().requires_grad_(True)
    y_ref = F.silu(gr[:, :I]) * gr[:, I:]
    y_ref.backward(dy.float())
    close(ops.swiglu(gu.to(dev), I), y_ref, 6e-3, "swiglu")
    close(ops.swiglu_bwd(gu.to(dev), dy.to(dev), I), gr.grad, 8e-3, "swiglu bwd")
    for act in (ops.ACT_RELU, ops.ACT_GELU, ops.ACT_QUICKGELU, ops.ACT_SILU):
        pre = rnd(rows, I, seed=35 + act)
        pr = pre.float().requires_grad_(True)
        act_ref(pr, act).backward(dy.float())
        close(ops.act_bwd(pre.to(dev), dy.to(dev), act), pr.grad, 8e-3, f"act_bwd {act}")
    a, b = rnd(rows, I, seed=40), rnd(rows, I, seed=41)
    close(ops.add(a.to(dev), b.to(dev)), a.float() + b.float(), 6e-3, "add")
    pe = rnd(7, I, seed=42)
    close(ops.add_bcast_rows(a.to(dev), pe.to(dev), 7), a.float() + pe.float()[torch.arange(rows) % 7], 6e-3, "add bcast")


def test_rows_scatter_colsum_cast(dev):
    from grove_amd import ops
    src = rnd(30, 64, seed=43)
    idx_s = torch.tensor([3, -1, 5, 29, 0, 0, 7], dtype=torch.int32)
    idx_d = torch.tensor([0, 1, 2, 3, 9, -1, 4], dtype=torch.int32)
    dst = torch.ones(10, 64, dtype=bf16, device=dev)
    ops.copy_rows(src.to(dev), dst, 7, 64, idx_src=idx_s.to(dev), idx_dst=idx_d.to(dev))
    ref = torch.ones(10, 64)
    for s_, d_ in zip(idx_s.tolist(), idx_d.tolist()):
        if d_ >= 0:
            ref[d_] = src[s_].float() if s_ >= 0 else 0
    close(dst, ref, 1e-7, "copy_rows")
    acc = torch.zeros(5, 64, dtype=torch.float32, device=dev)
    idx = (torch.arange(30) % 5).int()
    ops.scatter_add_f32(src.to(dev), acc, idx.to(dev), 30, 64)
    close(acc, torch.zeros(5, 64).index_add_(0, idx.long(), src.float()), 1e-5, "scatter_add")
    x = rnd(777, 130, seed=44)
    close(ops.colsum(x.to(dev)), x.float().sum(0), 1e-4, "colsum")
    f = torch.randn(1003, generator=torch.Generator().manual_seed(45))
    close(ops.to_bf16(f.to(dev)), f.to(bf16).float(), 1e-7, "cast f2b")
    close(ops.to_f32(f.to(bf16).to(dev)), f.to(bf16).float(), 1e-7, "cast b2f")


def test_im2col_and_pool(dev):
    from grove_amd import ops
    B, C, T, H, W, P = 1, 3, 2, 28, 42, 14
    img = rnd(B, C, T, H, W, seed=46)
    col = ops.im2col_patch(img.to(dev), P, 608)
    ref = img.float().permute(0, 2, 1, 3, 4).reshape(B * T, C, H // P, P, W // P, P).permute(0, 2, 4, 1, 3, 5).reshape(-1, C * P * P)
    close(col[:, :C * P * P], ref, 1e-7, "im2col")
    assert (col[:, C * P * P:].float() == 0).all()
    G, Cc = 2, 64
    x = rnd(G * 8, 577, Cc, seed=47)
    xr = x.float()[:, 1:].reshape(G, 8, 24, 24, Cc).permute(0, 4, 1, 2, 3)
    pref = F.adaptive_avg_pool3d(xr, (8, 8, 9)).permute(0, 2, 3, 4, 1).reshape(G, 576, Cc)
    close(ops.clip_pool(x.to(dev), G), pref, 6e-3, "clip_pool")


def test_cross_entropy(dev):
    from grove_amd import ops
    R, V, ld = 9, 1003, 1008
    logits = torch.zeros(R, ld, dtype=bf16)
    logits[:, :V] = rnd(R, V, seed=48, scale=3)
    labels = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(49))
    lr = logits[:, :V].float().requires_grad_(True)
    loss_ref = F.cross_entropy(lr, labels, reduction="sum")
    (loss_ref * 0.1).backward()
    dl = torch.zeros(R, ld, dtype=bf16, device=dev)
    gs = torch.tensor([0.1], device=dev)
    loss = ops.cross_entropy(logits.to(dev), labels.int().to(dev), V, dlogits=dl, grad_scale=gs)
    close(loss, loss_ref.reshape(1), 1e-4, "ce loss")
    close(dl[:, :V], lr.grad, 1e-2, "ce grad")


# ----------------------------------------------------------------------------- decoder kernels
@pytest.mark.parametrize("Lq,Lk,heads,d", [(6, 6, 8, 32), (8, 8, 3, 32), (3, 5, 5, 32), (1, 1, 1, 32), (6, 1024, 8, 16), (1024, 6, 8, 16), (5, 100, 2, 16),
                                           (8, 300, 4, 16), (3, 1500, 2, 16)])
def test_small_attn(dev, Lq, Lk, heads, d):
    from grove_amd import ops
    inst = 3
    q, k, v = rnd(inst * Lq, heads * d, seed=50), rnd(inst * Lk, heads * d, seed=51), rnd(inst * Lk, heads * d, seed=52)
    do = rnd(inst * Lq, heads * d, seed=53)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))

    def heads_(t, L):
        return t.reshape(inst, L, heads, d).transpose(1, 2)
    att = torch.softmax(heads_(qr, Lq) @ heads_(kr, Lk).transpose(-1, -2) / math.sqrt(d), -1)
    o_ref = (att @ heads_(vr, Lk)).transpose(1, 2).reshape(inst * Lq, heads * d)
    o_ref.backward(do.float())
    o = ops.small_attn(q.to(dev), k.to(dev), v.to(dev), inst, heads, d, Lq, Lk)
    close(o, o_ref, 8e-3, "small attn fwd")
    dq, dk, dv = ops.small_attn_bwd(q.to(dev), k.to(dev), v.to(dev), o, do.to(dev), inst, heads, d, Lq, Lk)
    close(dq, qr.grad, 2e-2, "dq")
    close(dk, kr.grad, 2e-2, "dk")
    close(dv, vr.grad, 2e-2, "dv")


@pytest.mark.parametrize("Lq,Lk,heads", [(6, 6, 8), (8, 3, 3), (2, 8, 5)])
def test_small_attn_tiny_kernels_match_the_generic_ones(dev, Lq, Lk, heads):
    """Round 6b: Lq, Lk <= 8 at head dim 32 (the box decoder's token self attention, transformer.py:153-160) on the lane-per-(pair, row)
    kernels against the generic few-keys kernels (grove_small_attn_set_tiny(0)) on the same operands: a partial last wave of pairs,
    gradients stored once (NaN-filled outputs must come back fully written, and identical from run to run)."""
    from grove_amd import _lib, ops
    inst, d = 13, 32
    q, k, v = rnd(inst * Lq, heads * d, seed=70), rnd(inst * Lk, heads * d, seed=71), rnd(inst * Lk, heads * d, seed=72)
    do = rnd(inst * Lq, heads * d, seed=73)
    args = (q.to(dev), k.to(dev), v.to(dev))
    L = _lib.lib()
    try:
        L.grove_small_attn_set_tiny(0)
        o_g = ops.small_attn(*args, inst, heads, d, Lq, Lk)
        g_g = ops.small_attn_bwd(*args, o_g, do.to(dev), inst, heads, d, Lq, Lk)
    finally:
        L.grove_small_attn_set_tiny(1)
    o_t = ops.small_attn(*args, inst, heads, d, Lq, Lk)
    g_t = ops.small_attn_bwd(*args, o_t, do.to(dev), inst, heads, d, Lq, Lk)
    g_t2 = ops.small_attn_bwd(*args, o_t, do.to(dev), inst, heads, d, Lq, Lk)
    close(o_t, o_g, 4e-3, "o")
    for a, b, c, name in zip(g_t, g_g, g_t2, ("dq", "dk", "dv")):
        assert torch.isfinite(a).all(), name
        close(a, b, 1e-4, name)
        assert torch.equal(a, c), name + ": no atomics, so two runs are the same bits"


@pytest.mark.parametrize("Lq,Lk,heads,d", [(6, 6, 8, 32), (6, 1024, 8, 16), (1024, 6, 8, 16), (5, 100, 2, 16)])
def test_small_attn_bwd_bf16_gradients(dev, Lq, Lk, heads, d):
    """grove_small_attn_params.grad_bf16: the kernels that store every gradient element once write bf16 directly — the same values as the
    fp32 arrays rounded afterwards; the other families return fp32 and ops casts (same result either way)."""
    from grove_amd import ops
    inst = 5
    q, k, v = rnd(inst * Lq, heads * d, seed=90), rnd(inst * Lk, heads * d, seed=91), rnd(inst * Lk, heads * d, seed=92)
    do = rnd(inst * Lq, heads * d, seed=93)
    args = (q.to(dev), k.to(dev), v.to(dev))
    o = ops.small_attn(*args, inst, heads, d, Lq, Lk)
    g32 = ops.small_attn_bwd(*args, o, do.to(dev), inst, heads, d, Lq, Lk)
    g16 = ops.small_attn_bwd(*args, o, do.to(dev), inst, heads, d, Lq, Lk, bf16_grads=True)
    for a, b, name in zip(g16, g32, ("dq", "dk", "dv")):
        assert a.dtype == bf16 and b.dtype == torch.float32
        if Lk > 8 or (Lq <= 8 and d == 32):   # stored once: deterministic, so exactly the rounding of the fp32 run
            assert torch.equal(a, b.to(bf16)), name
        else:                                 # the atomics families: two runs differ by accumulation order
            close(a, b, 1e-2, name)


def test_transpose_many(dev):
    """grove_transpose_many: a list of small matrices transposed in one launch — ragged tile edges, a strided source, and the cache's
    promise that a second call REWRITES the same outputs from the current values."""
    from grove_amd import ops
    shapes = [(256, 256), (128, 256), (256, 128), (2048, 256), (72, 40), (8, 8), (65, 129)]
    ws = [rnd(r, c, seed=100 + i).to(dev) for i, (r, c) in enumerate(shapes)]
    wide = rnd(96, 200, seed=120).to(dev)
    ws.append(wide[:, 8:136])  # row stride 200, 128 columns
    outs = ops.transpose2d_many(ws)
    for w, o in zip(ws, outs):
        assert o.shape == (w.shape[1], w.shape[0]) and torch.equal(o, w.t().contiguous())
    ws[0].mul_(2)
    outs2 = ops.transpose2d_many(ws)
    assert outs2[0].data_ptr() == outs[0].data_ptr() and torch.equal(outs2[0], ws[0].t().contiguous())


def test_segment_sum_rows_matches_scatter_add(dev):
    """Round 6b: grove_segment_sum_rows (one owner per output element) against the atomics form on the same grouped index: empty segments,
    segments of 1..4 members, an accumulate into a non-zero destination; exact against an fp32 sum in member order, and the same bits twice."""
    from grove_amd import ops
    R, Cc = 48, 64
    counts = [3, 0, 1, 4, 2, 0, 3]
    ptr = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
    n = int(ptr[-1])
    src = rnd(n * R, Cc, seed=81)
    dst0 = torch.randn(len(counts) * R, Cc, generator=torch.Generator().manual_seed(82))
    ref = dst0.clone()
    for s_, (lo, hi) in enumerate(zip(ptr[:-1].tolist(), ptr[1:].tolist())):
        for i in range(lo, hi):
            ref[s_ * R:(s_ + 1) * R] += src[i * R:(i + 1) * R].float()
    a = ops.segment_sum_rows(src.to(dev), dst0.clone().to(dev), ptr.to(dev), R)
    b = ops.segment_sum_rows(src.to(dev), dst0.clone().to(dev), ptr.to(dev), R)
    seg_of = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts))
    idx = (seg_of[:, None] * R + torch.arange(R)[None]).reshape(-1).to(torch.int32)
    c = ops.scatter_add_f32(src.to(dev), dst0.clone().to(dev), idx.to(dev), n * R, Cc)
    assert torch.equal(a, b)
    close(a, ref, 1e-6, "segment sum vs fp32")
    close(a, c, 1e-6, "segment sum vs scatter-add")


def test_box_head_and_losses(dev):
    from grove_amd import ops
    N, D = 11, 256
    g = torch.Generator().manual_seed(54)
    x = torch.randn(N, D, generator=g)
    W1, b1, W2, b2, Wo, bo = (rnd(D, D, seed=55, scale=0.05), rnd(D, seed=56, scale=0.1), rnd(4, D, seed=57, scale=0.05),
                              rnd(4, seed=58, scale=0.1), rnd(1, D, seed=59, scale=0.05), rnd(1, seed=60))
    xr = x.clone().requires_grad_(True)
    params = [t.float().requires_grad_(True) for t in (W1, b1, W2, b2, Wo, bo)]
    h_ref = torch.relu(xr @ params[0].t() + params[1])
    box_ref = torch.sigmoid(h_ref @ params[2].t() + params[3])
    obj_ref = (xr @ params[4].t() + params[5]).squeeze(-1)
    dW = [t.to(dev) for t in (W1, b1, W2, b2, Wo, bo)]
    box, obj, hidden = ops.box_head(x.to(dev), *dW)
    close(box, box_ref, 1e-5, "box")
    close(obj, obj_ref, 1e-5, "obj")
    # losses vs autograd of the published torchvision GIoU formula
    gt = torch.rand(N, 4, generator=g) * 0.5 + 0.2
    vis = (torch.rand(N, generator=g) > 0.3).float()
    vis[0] = 1

    def to_xyxy(b):
        cx, cy, w, h = b.unbind(-1)
        return torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)

    def giou_sum(b1_, b2_, eps=1e-7):
        x1, y1, x2, y2 = b1_.unbind(-1)
        x1g, y1g, x2g, y2g = b2_.unbind(-1)
        xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
        inter = torch.where((yk2 > yk1) & (xk2 > xk1), (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
        union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
        iou = inter / (union + eps)
        xc1, yc1, xc2, yc2 = torch.min(x1, x1g), torch.min(y1, y1g), torch.max(x2, x2g), torch.max(y2, y2g)
        ac = (xc2 - xc1) * (yc2 - yc1)
        return (1 - (iou - (ac - union) / (ac + eps))).sum()
    m = vis.bool()
    ngt = float(m.sum())
    gi = giou_sum(to_xyxy(box_ref[m]), to_xyxy(gt[m]))
    l1 = (box_ref[m] - gt[m]).abs().sum()
    bce = F.binary_cross_entropy_with_logits(obj_ref, vis, reduction="sum")
    total = 2.0 * (gi + l1) / ngt + 0.5 * bce / N
    total.backward()
    sums, dbox, dobj = ops.box_losses(box, obj, gt.to(dev), vis.to(dev), 2.0 / ngt, 0.5 / N)
    close(sums, torch.stack([gi, l1, bce]), 1e-4, "loss sums")
    grads = {k_: torch.zeros(s_, dtype=torch.float32, device=dev) for k_, s_ in
             dict(dW1=(D, D), db1=(D,), dW2=(4, D), db2=(4,), dWo=(D,), dbo=(1,)).items()}
    dx = ops.box_head_bwd(x.to(dev), dW[0], dW[2], dW[4], hidden, box, dbox, dobj, grads)
    close(dx, xr.grad, 1e-3, "head dx")
    for name, pr in zip(["dW1", "db1", "dW2", "db2", "dWo", "dbo"], params):
        close(grads[name].reshape(pr.grad.shape), pr.grad, 1e-3, name)


def test_adamw_sumsq(dev):
    from grove_amd import ops
    n = 1000
    g = torch.Generator().manual_seed(61)
    w, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.AdamW([p], lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    master, m, v = w.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    model = torch.empty(n, dtype=bf16, device=dev)
    for step in (1, 2, 3):
        p.grad = gr * 0.5
        opt.step()
        ops.adamw_step(master, model, gr.to(dev), m, v, 3e-4, 0.9, 0.95, 1e-8, 0.01, 0.5, step)
    close(master, p.detach(), 1e-5, "adamw")
    close(model, p.detach().to(bf16), 1e-7 + 4e-3, "adamw bf16 copy")
    close(ops.sumsq(gr.to(dev)), (gr * gr).sum().reshape(1), 1e-5, "sumsq")


def test_gemm_split_k_atomics(dev):
    """Accumulating f32 GEMMs (weight gradients) with K split over blockIdx.z, partials met by fp32 atomics."""
    from grove_amd import ops
    M, N, K = 128, 256, 32 * 96
    a, b = rnd(M, K, seed=70), rnd(N, K, seed=71)
    c0 = torch.randn(M, N, generator=torch.Generator().manual_seed(72))
    ref = c0 + a.float() @ b.float().t()
    for split in (0, 1, 5, 24):
        out = c0.clone().to(dev)
        ops.gemm_raw(a.to(dev), b.to(dev), out, M, N, K, K, K, N, accumulate=True, split_k=split)
        close(out, ref, 3e-5, f"split_k={split}")


@pytest.mark.parametrize("L", [1024, 1000, 130])
def test_rel_pos_indicator_in_registers_is_bit_identical(dev, L):
    """Round 4: SAM's global blocks (rel_kw = rel_kh = 32, rel_ld = 64, head dim 96) take kernel instances whose rel-pos indicator
    fragments are made in registers instead of an LDS tile rebuilt per key tile — the same MFMAs on the same operand values: the forward (the
    only kernel that gained from it; the backward runs with both settings here too) is bit-identical to the LDS form; ragged last tiles included."""
    from grove_amd import _lib, ops
    Lb = _lib.lib()
    B, H, hs, hd = 2, 3, 96, 80
    g = torch.Generator().manual_seed(L)
    qkv = torch.zeros(B * L, 3 * H * hs)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g)
    qkv = qkv.to(bf16).to(dev)
    do = (torch.randn(B * L, H * hs, generator=g)).to(bf16).to(dev)
    rel = (torch.randn(B * H, L, 64, generator=g) / hd ** -0.5).to(bf16).to(dev)
    res = []
    try:
        for on in (0, 1):
            Lb.grove_flash_attn_set_register_e(on)
            out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, rel=rel, rel_hw=(32, 32), want_lse=True, hs_valid=hd)
            dqkv = torch.full_like(qkv, float("nan"))
            drel = ops.flash_attn_bwd(qkv, out, do, lse, dqkv, B, L, H, hs, 0, H * hs, 2 * H * hs, hd ** -0.5, rel=rel, rel_hw=(32, 32), want_drel=True,
                                      hs_valid=hd)
            res.append((out, lse, dqkv, drel))
    finally:
        Lb.grove_flash_attn_set_register_e(1)
    for a, b, name in zip(res[0], res[1], ("o", "lse", "dqkv", "drel")):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("B,H,L,hs,hd,causal,use_len,rel_hw", [
    (2, 3, 77, 64, 64, False, False, None),
    (2, 2, 200, 128, 128, True, True, None),
    (3, 2, 196, 96, 80, False, False, (14, 14)),
    (1, 2, 1024, 96, 80, False, False, (32, 32)),
    (2, 4, 50, 32, 16, False, False, (5, 10)),
    (1, 2, 703, 128, 128, True, False, None),
    (5, 16, 196, 96, 80, False, False, (14, 14)),   # SAM window problem, many (window, head) pairs: the LDS-resident window kernels
    (2, 3, 208, 96, 80, False, False, (13, 16)),    # every key / query tile full
    (2, 2, 195, 96, 80, False, False, (13, 15)),    # three valid rows in the last tile
])
def test_flash_attention_fwd_bwd(dev, B, H, L, hs, hd, causal, use_len, rel_hw):
    """Fused attention (fwd, dQ/dK/dV, d rel) against torch autograd in fp32 on the same bf16 inputs.
    hd < hs exercises the zero-padded head layout used for SAM (80 -> 96)."""
    from grove_amd import ops
    g = torch.Generator().manual_seed(80)
    qkv = torch.zeros(B * L, 3 * H * hs)
    qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g)
    qkv = qkv.to(bf16)
    do = torch.zeros(B * L, H * hs)
    do.view(B * L, H, hs)[..., :hd] = torch.randn(B * L, H, hd, generator=g)
    do = do.to(bf16)
    alpha = hd ** -0.5
    kv_len = torch.tensor([L, max(1, L - 37)][:B] + [L] * max(0, B - 2), dtype=torch.int32) if use_len else None
    rel = relp = None
    if rel_hw is not None:
        kh, kw = rel_hw
        khp = (kh + 15) // 16 * 16
        rel_ld = khp + (kw + 15) // 16 * 16
        # the kernel consumes rel' = rel / alpha in bf16, h-bins at 0.., w-bins at khp..
        relp = torch.zeros(B * H, L, rel_ld)
        relp[..., :kh] = torch.randn(B * H, L, kh, generator=g) / alpha
        relp[..., khp:khp + kw] = torch.randn(B * H, L, kw, generator=g) / alpha
        relp = relp.to(bf16)
    t = qkv.float().view(B, L, 3, H, hs).requires_grad_(True)
    q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)  # [B,H,L,hs]
    s = q @ k.transpose(-1, -2) * alpha
    relr = None
    if relp is not None:
        relr = relp.float().clone().requires_grad_(True)
        bias = (relr[..., :kh, None] + relr[..., None, khp:khp + kw]) * alpha
        s = s + bias.reshape(B, H, L, L)
    mask = torch.zeros(B, 1, L, L, dtype=torch.bool)
    if causal:
        mask |= torch.ones(L, L, dtype=torch.bool).triu(1)[None, None]
    if kv_len is not None:
        for b in range(B):
            mask[b, :, :, kv_len[b]:] = True
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1)
    o_ref = (p @ v).transpose(1, 2).reshape(B * L, H * hs)
    o_ref.backward(do.float())
    lse_ref = torch.logsumexp(s, -1).reshape(B * H, L)
    dev_qkv = qkv.to(dev)
    rel_d = relp.to(dev) if relp is not None else None
    rel_arg = (khp, kw) if relp is not None else (0, 0)
    kvl = kv_len.to(dev) if kv_len is not None else None
    hv = hd if hd < hs else 0
    out, lse = ops.flash_attn(dev_qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=causal, kv_len=kvl, rel=rel_d,
                              rel_hw=rel_arg, want_lse=True, hs_valid=hv)
    close(out, o_ref, 1e-2, "flash fwd")
    close(lse, lse_ref, 2e-3, "lse")
    if hv == 80 and 192 < L <= 208 and rel_arg[0] + 16 == 32:
        # this problem ran on the window kernels (win_attn.hip): same result from the general kernels, pad columns exact zeros
        from grove_amd import _lib
        _lib.lib().grove_flash_attn_set_window_kernels(0)
        try:
            out_g, lse_g = ops.flash_attn(dev_qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel_d, rel_hw=rel_arg, want_lse=True,
                                          hs_valid=hv)
        finally:
            _lib.lib().grove_flash_attn_set_window_kernels(1)
        close(out, out_g.float().cpu(), 1e-2, "window kernel vs general kernel")
        close(lse, lse_g.float().cpu(), 2e-3, "window kernel lse vs general kernel")
        assert float(out.view(B * L, H, hs)[..., hd:].float().abs().max()) == 0.0
    dqkv = torch.full_like(dev_qkv, float("nan"))
    drel = ops.flash_attn_bwd(dev_qkv, out, do.to(dev), lse, dqkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=causal,
                              kv_len=kvl, rel=rel_d, rel_hw=rel_arg, want_drel=relp is not None, hs_valid=hv)
    gref = t.grad.reshape(B * L, 3 * H * hs)
    for name, c0 in (("dq", 0), ("dk", H * hs), ("dv", 2 * H * hs)):
        close(dqkv[:, c0:c0 + H * hs], gref[:, c0:c0 + H * hs], 2e-2, name)
    if relp is not None:
        close(drel, relr.grad, 2e-2, "drel")


@pytest.mark.parametrize("K,M,N,split", [(100, 64, 128, 0), (3000, 256, 128, 0), (777, 128, 256, 7), (64, 8, 8, 1)])
def test_wgrad_tn_gemm(dev, K, M, N, split):
    """dW += dY^T X on K-major operands (no transposed copies), incl. K tails and split-K atomics."""
    from grove_amd import ops
    dy, x = rnd(K, M, seed=90), rnd(K, N, seed=91)
    g0 = torch.randn(M, N, generator=torch.Generator().manual_seed(92))
    out = g0.clone().to(dev)
    ops.wgrad(dy.to(dev), x.to(dev), out, split_k=split, alpha=0.5)
    close(out, g0 + 0.5 * dy.float().t() @ x.float(), 3e-5, "wgrad")


def test_wgrad_conv3d_gather(dev):
    """Conv3d weight gradient as ONE gathered TN GEMM, against autograd of F.conv3d."""
    from grove_amd import ops
    from grove_amd.model.indexing import conv3d_gather_index
    G, T, H, W, Ci, Co = 1, 2, 5, 6, 128, 64
    x = rnd(G * T * H * W, Ci, seed=93)
    dy = rnd(G * T * H * W, Co, seed=94)
    w = torch.zeros(Co, Ci, 3, 3, 3, requires_grad=True)
    xr = x.float().reshape(G, T, H, W, Ci).permute(0, 4, 1, 2, 3)
    y = F.conv3d(xr, w, padding=1).permute(0, 2, 3, 4, 1).reshape(-1, Co)
    y.backward(dy.float())
    ref = w.grad.permute(0, 2, 3, 4, 1).reshape(Co, 27 * Ci)  # tap-major
    idx = conv3d_gather_index(G, T, H, W).to(dev)
    out = torch.zeros(Co, 27 * Ci, dtype=torch.float32, device=dev)
    ops.wgrad(dy.to(dev), x.to(dev), out, b_idx=idx, b_taps=27)
    close(out, ref, 3e-5, "conv3d wgrad")


@pytest.mark.parametrize("tile_n,tile_m", [(64, 128), (128, 128), (128, 192)])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (2812, 520, 128), (130, 64, 64)])
def test_gemm_tile_variants(dev, M, N, K, tile_n, tile_m):
    """All macro tiles (128 x 128, 192 x 128, 128 x 64) with a full epilogue."""
    from grove_amd import ops, _lib
    ops.gemm_set_tile_n(tile_n)
    _lib.lib().grove_gemm_set_tile_m(tile_m)
    try:
        a, b = rnd(M, K, seed=95), rnd(N, K, seed=96, scale=0.1)
        bias, res = rnd(N, seed=97), rnd(M, N, seed=98)
        ref = F.gelu(a.float() @ b.float().t() + bias.float()) + res.float()
        out = ops.linear(a.to(dev), b.to(dev), bias.to(dev), act=ops.ACT_GELU, residual=res.to(dev))
        close(out, ref, 8e-3, f"tile_n={tile_n} tile_m={tile_m}")
    finally:
        ops.gemm_set_tile_n(0)
        _lib.lib().grove_gemm_set_tile_m(0)


@pytest.mark.parametrize("M,N1,K1,N2", [(8, 4096, 4096, 22016), (3, 512, 11008, 1536), (1, 4096, 4096, 32008)])
def test_gemv_deferred_rmsnorm(dev, M, N1, K1, N2):
    """The deferred RMSNorm of the batched decode step (grove_gemv_params.xs_out / ssq_out / ssq_in): a producer GEMV (o_proj / down_proj
    shape: fp32 residual in, fp32 stream out) that also leaves bf16(stream * next norm weight) and its per-16-column sums of squares, and
    a consumer GEMV (gate | up with the SwiGLU epilogue, or lm_head) that scales its product by the row's rstd — against
    rmsnorm(stream) * weight -> GEMV in fp32, and bit-identical between M sequences and the same rows one at a time (force_mfma)."""
    from grove_amd import ops
    g = torch.Generator().manual_seed(21)
    o = (torch.randn(M, K1, generator=g) * 0.5).to(bf16)
    w1 = (torch.randn(N1, K1, generator=g) * 0.02).to(bf16)
    res = torch.randn(M, N1, generator=g)
    nw = (1.0 + 0.1 * torch.randn(N1, generator=g)).to(bf16)
    swiglu = N2 % 16 == 0 and N2 != 32008
    w2 = (torch.randn(N2, N1, generator=g) * 0.02).to(bf16)
    y, xs, ssq = ops.gemv(o.to(dev), w1.to(dev), residual=res.to(dev), out_dtype=torch.float32, batch_invariant=True, norm_out=nw.to(dev))
    y_ref = o.float() @ w1.float().t() + res
    close(y, y_ref, 2e-5, "producer: the stream")
    close(xs, y.float().cpu() * nw.float(), 2 ** -8, "producer: bf16(stream * weight)")
    assert ssq.shape == ((N1 + 15) // 16, 8)
    close(ssq.cpu().sum(0)[:M], y.float().cpu().pow(2).sum(1), 1e-5, "producer: sums of squares")
    eps = 1e-5
    act = ops.ACT_SWIGLU_PAIR if swiglu else ops.ACT_NONE
    out = ops.gemv(xs, w2.to(dev), act=act, out_dtype=torch.float32 if not swiglu else bf16, batch_invariant=True, norm_in=(ssq, eps))
    yn = y.float().cpu()
    xn = (yn * torch.rsqrt(yn.pow(2).mean(1, keepdim=True) + eps) * nw.float())
    z = xn @ w2.float().t()
    if swiglu:  # rows interleaved 4 gate / 4 up per 8
        z8 = z.view(M, N2 // 8, 2, 4)
        want = (torch.nn.functional.silu(z8[:, :, 0]) * z8[:, :, 1]).reshape(M, N2 // 2)
        close(out, want, 1.5e-2, "consumer: SwiGLU of the normalised product")
    else:
        close(out, z, 6e-3, "consumer: the normalised product")
    # batch invariance of the pair: row 0 alone gives the same bits
    y1, xs1, ssq1 = ops.gemv(o[:1].to(dev), w1.to(dev), residual=res[:1].to(dev), out_dtype=torch.float32, batch_invariant=True, norm_out=nw.to(dev))
    out1 = ops.gemv(xs1, w2.to(dev), act=act, out_dtype=out.dtype, batch_invariant=True, norm_in=(ssq1, eps))
    assert torch.equal(y1[0], y[0]) and torch.equal(xs1[0], xs[0]) and torch.equal(ssq1[:, 0], ssq[:, 0]) and torch.equal(out1[0], out[0])


@pytest.mark.parametrize("M", [3, 5, 6, 7])
def test_gemv_touches_only_its_m_rows(dev, M):
    """3, 5, 6 and 7 sequences run the 4- / 8-row instances of the VALU kernel (K % 128 != 0 keeps them off the matrix-core one): x,
    the residual and y have M rows, not 4 / 8. Round 5 found the epilogue storing (and reading the residual of) all MX rows — past the
    end of y, a fault only when y ends a mapped segment. x, residual and y are the LAST rows of larger buffers here: the rows behind
    them must keep their guard values, and the result must not depend on them."""
    from grove_amd import ops
    N, K = 520, 1096
    w = rnd(N, K, seed=2, scale=0.05).to(dev)
    xb = torch.full((M + 8, K), float("nan"), dtype=bf16, device=dev)
    rb = torch.full((M + 8, N), float("nan"), dtype=bf16, device=dev)
    yb = torch.full((M + 8, N), 7.0, dtype=bf16, device=dev)
    x, res = rnd(M, K, seed=1), rnd(M, N, seed=4)
    xb[:M], rb[:M] = x.to(dev), res.to(dev)
    ops.gemv(xb[:M], w, residual=rb[:M], out=yb[:M])
    torch.cuda.synchronize()
    assert torch.equal(yb[M:], torch.full((8, N), 7.0, dtype=bf16, device=dev)), "rows behind y were written"
    close(yb[:M], x.float() @ w.float().cpu().t() + res.float(), 2 ** -7, "gemv M rows")


@pytest.mark.parametrize("M,N,K", [(1, 300, 512), (2, 4096, 4096), (3, 1000, 1096), (4, 515, 11008), (7, 64, 256), (8, 320, 128)])
def test_gemv_decode(dev, M, N, K):
    """grove_gemv_bf16 (cached decode step) vs fp32 reference, with bias / activation / residual / fp32 output."""
    from grove_amd import ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = x.float() @ w.float().t()
    y = ops.gemv(x.to(dev), w.to(dev))
    close(y, ref, 2 ** -7, "gemv plain")
    y = ops.gemv(x.to(dev), w.to(dev), bias.to(dev), act=ops.ACT_RELU, residual=res.to(dev))
    close(y, F.relu(ref + bias.float()) + res.float(), 2 ** -7, "gemv bias+relu+residual")
    y = ops.gemv(x.to(dev), w.to(dev), out_dtype=torch.float32)
    close(y, ref, 1e-5, "gemv f32")
    # the tile GEMM on the same operands: both accumulate in fp32, results agree to bf16 rounding
    close(y, ops.linear(x.to(dev), w.to(dev), out_dtype=torch.float32) if K % 32 == 0 else ref, 1e-5, "gemv vs gemm")


@pytest.mark.parametrize("B,H,hd,Lk,Smax", [(2, 4, 32, 37, 64), (1, 8, 128, 700, 768), (3, 2, 64, 1, 16)])
def test_flash_attention_decode_against_cache(dev, B, H, hd, Lk, Smax):
    """One query row per sequence against a [B, S_max, 2H] key|value cache (LlamaStack.decode_step)."""
    from grove_amd import ops
    Hd = H * hd
    qkv = rnd(B, 3 * Hd, seed=5)
    kv = rnd(B, Smax, 2 * Hd, seed=6)
    o = ops.flash_attn_kv(qkv.to(dev), kv.to(dev), kv.to(dev)[:, :, Hd:], B, H, 1, Lk, hd, hd ** -0.5, sq=3 * Hd, sk=Smax * 2 * Hd,
                          sv=Smax * 2 * Hd, ld_q=3 * Hd, ld_k=2 * Hd, ld_v=2 * Hd)
    q = qkv[:, :Hd].float().view(B, H, 1, hd)
    k = kv[:, :Lk, :Hd].float().view(B, Lk, H, hd).permute(0, 2, 1, 3)
    v = kv[:, :Lk, Hd:].float().view(B, Lk, H, hd).permute(0, 2, 1, 3)
    ref = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
    close(o.view(B, H, hd), ref.view(B, H, hd), 2 ** -7, "decode attention")


def test_gemv_fused_rmsnorm_and_swiglu(dev):
    """The decode step's folded input transforms: rmsnorm(x)*w and silu(gate)*up, against the standalone kernels."""
    from grove_amd import ops
    M, K, N = 2, 1024, 520
    x, w, nw = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2, scale=0.05).to(dev), rnd(K, seed=3).to(dev)
    nw0 = nw
    y = ops.gemv(x, w, rms_weight=nw, eps=1e-5)
    y_ref = ops.gemv(ops.rmsnorm(x, nw, 1e-5), w)
    close(y, y_ref, 2 ** -8, "gemv rmsnorm-fused vs rmsnorm kernel + gemv")
    xf = x.float().cpu()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * nw.float().cpu()).to(bf16).float() @ w.float().cpu().t()
    close(y, ref, 2 ** -7, "gemv rmsnorm-fused vs fp32")
    gu = rnd(M, 2 * K, seed=4).to(dev)
    y = ops.gemv(gu, w, swiglu=True)
    close(y, ops.gemv(ops.swiglu(gu, K), w), 2 ** -8, "gemv swiglu-fused vs swiglu kernel + gemv")
    # SwiGLU in the EPILOGUE of the gate|up GEMV (rows interleaved 4 gate / 4 up; round 3): bit-identical to GEMV + swiglu kernel
    I, Kx = 96, 2304  # (K >= 2048: the weight trip that is issued before the x prologue, plus a ragged tail)
    wgu = rnd(2 * I, Kx, seed=21, scale=0.1).to(dev)
    xs = rnd(M, Kx, seed=22).to(dev)
    nw = rnd(Kx, seed=23).to(dev)
    ref = ops.swiglu(ops.gemv(xs, wgu, rms_weight=nw, eps=1e-5), I)
    got = ops.gemv(xs, ops.swiglu_interleave(wgu), rms_weight=nw, eps=1e-5, act=ops.ACT_SWIGLU_PAIR)
    assert got.shape == (M, I) and torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    # the decode step's FP32 residual stream (round 3): x fp32 with the norm folded (statistics on the fp32 values), fp32 residual, fp32 out
    xf32 = (x.float() * 1.001).contiguous()                      # values that are NOT bf16-representable
    res32 = rnd(M, N, seed=31).to(dev).float() * 1.003
    y32 = ops.gemv(xf32, w, rms_weight=nw0, eps=1e-5, residual=res32, out_dtype=torch.float32)
    xc = xf32.cpu()
    refn = (xc * torch.rsqrt(xc.pow(2).mean(-1, keepdim=True) + 1e-5) * nw0.float().cpu()).to(bf16).float()
    ref32 = refn @ w.float().cpu().t() + res32.cpu()
    assert y32.dtype == torch.float32
    close(y32, ref32, 2 ** -10, "gemv with fp32 x (norm folded) + fp32 residual vs fp32")
    yp = ops.gemv(xf32, w, out_dtype=torch.float32)               # plain fp32 x: rounded to bf16 for the products
    close(yp, xc.to(bf16).float() @ w.float().cpu().t(), 2 ** -10, "gemv with plain fp32 x")


@pytest.mark.parametrize("M", [3, 4, 5, 8])
def test_gemv_matrix_core_kernel_for_batched_decode(dev, M):
    """Round 5: 3..8 sequences (the clip-batched decode) run the weight stream through v_mfma_f32_16x16x32_bf16 (gemv_mfma_kernel):
    every prologue (plain x from global memory incl. LLaMA's K = 11008 that does not fit the LDS at 8 rows, folded RMSNorm on a bf16
    and on an fp32 stream, fused SwiGLU input) and epilogue (bias + activation + residual, fp32 output, the SwiGLU pair) against
    the VALU kernel of the same library (grove_gemv_set_mfma(0); one row at a time where 8 rows of K do not fit its LDS) and fp32."""
    from grove_amd import _lib, ops
    Lb = _lib.lib()

    def both(fn):
        Lb.grove_gemv_set_mfma(1)
        a = fn()
        Lb.grove_gemv_set_mfma(0)
        try:
            b = fn()
        finally:
            Lb.grove_gemv_set_mfma(1)
        return a, b
    # plain x, long K (the down projection), ragged N
    K, N = 11008, 200
    x, w = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2, scale=0.05).to(dev)
    bias, res = rnd(N, seed=3).to(dev), rnd(M, N, seed=4).to(dev)
    y = ops.gemv(x, w, bias, act=ops.ACT_RELU, residual=res)
    ref = F.relu(x.float().cpu() @ w.float().cpu().t() + bias.float().cpu()) + res.float().cpu()
    close(y, ref, 2 ** -7, "mfma gemv plain K=11008 vs fp32")
    Lb.grove_gemv_set_mfma(0)
    try:
        y1 = torch.cat([ops.gemv(x[i:i + 1], w, bias, act=ops.ACT_RELU, residual=res[i:i + 1]) for i in range(M)], 0)
    finally:
        Lb.grove_gemv_set_mfma(1)
    close(y, y1.float().cpu(), 2 ** -7, "mfma gemv vs VALU kernel row by row")
    # wide projection (N >= 16384): two 16-row groups per workgroup share the x fragments; ragged N (last workgroup: one group, 8 rows);
    # bit-identical to the 16-row form (grove_gemv_set_mfma(3)): same products, same sum order per output
    K, N = 256, 16400
    xw, ww = rnd(M, K, seed=11).to(dev), rnd(N, K, seed=12, scale=0.05).to(dev)
    bw, rw = rnd(N, seed=13).to(dev), rnd(M, N, seed=14).to(dev)
    y32 = ops.gemv(xw, ww, bw, act=ops.ACT_RELU, residual=rw)
    Lb.grove_gemv_set_mfma(3)
    try:
        y16 = ops.gemv(xw, ww, bw, act=ops.ACT_RELU, residual=rw)
    finally:
        Lb.grove_gemv_set_mfma(1)
    assert torch.equal(y32, y16)
    close(y32, F.relu(xw.float().cpu() @ ww.float().cpu().t() + bw.float().cpu()) + rw.float().cpu(), 2 ** -7, "mfma gemv 32-row workgroups vs fp32")
    # folded RMSNorm (bf16 x), K = 4096; fp32 output
    K, N = 4096, 528
    x, w, nw = rnd(M, K, seed=5).to(dev), rnd(N, K, seed=6, scale=0.05).to(dev), rnd(K, seed=7).to(dev)
    a, b = both(lambda: ops.gemv(x, w, rms_weight=nw, eps=1e-5, out_dtype=torch.float32))
    close(a, b.cpu(), 2 ** -9, "mfma gemv rmsnorm-folded vs VALU kernel")
    # fp32 stream input with the norm folded + fp32 residual (the inference model's decode step)
    xf = (x.float() * 1.001).contiguous()
    r32 = rnd(M, N, seed=8).to(dev).float() * 1.003
    a, b = both(lambda: ops.gemv(xf, w, rms_weight=nw, eps=1e-5, residual=r32, out_dtype=torch.float32))
    close(a, b.cpu(), 2 ** -9, "mfma gemv fp32-stream vs VALU kernel")
    # fused SwiGLU input (gate | up row)
    K2 = 1024
    gu, w2 = rnd(M, 2 * K2, seed=9).to(dev), rnd(N, K2, seed=10, scale=0.05).to(dev)
    a, b = both(lambda: ops.gemv(gu, w2, swiglu=True))
    close(a, b.float().cpu(), 2 ** -7, "mfma gemv swiglu-input vs VALU kernel")
    # the SwiGLU pair epilogue of the gate|up projection (rows interleaved 4 gate / 4 up)
    I, Kx = 96, 2304
    wgu, xs2, nw2 = rnd(2 * I, Kx, seed=21, scale=0.1).to(dev), rnd(M, Kx, seed=22).to(dev), rnd(Kx, seed=23).to(dev)
    a, b = both(lambda: ops.gemv(xs2, ops.swiglu_interleave(wgu), rms_weight=nw2, eps=1e-5, act=ops.ACT_SWIGLU_PAIR))
    assert a.shape == (M, I)
    close(a, b.float().cpu(), 2 ** -7, "mfma gemv swiglu-pair epilogue vs VALU kernel")


@pytest.mark.parametrize("B,H,hd,t,Smax", [(2, 4, 32, 36, 64), (1, 8, 128, 699, 768), (3, 2, 64, 0, 16), (2, 32, 128, 1500, 2048)])
def test_decode_attention_fused(dev, B, H, hd, t, Smax):
    """grove_decode_attn = rope(q, k) + cache append + one-query attention, vs the unfused kernels and fp32."""
    from grove_amd import ops
    Hd = H * hd
    qkv = rnd(B, 3 * Hd, seed=5).to(dev)
    cache = rnd(B, 2, H, Smax, hd, seed=6).to(dev)             # [B, keys / values, head, position, hd]: the KVCache layout
    pos = torch.full((B,), t, dtype=torch.int32, device=dev)
    ref_qkv, ref_cache = qkv.clone(), cache.clone()
    ops.rope_(ref_qkv, pos, 0, 2 * H, hd, 10000.0)
    ref_cache[:, :, :, t] = ref_qkv[:, Hd:].view(B, 2, H, hd)
    q = ref_qkv[:, :Hd].float().cpu().view(B, H, 1, hd)
    k = ref_cache[:, 0, :, :t + 1].float().cpu()               # [B, H, t + 1, hd]
    v = ref_cache[:, 1, :, :t + 1].float().cpu()
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).view(B, Hd)
    o = ops.decode_attn(qkv, cache, pos, H, hd, 10000.0, hd ** -0.5)
    close(o, ref, 2 ** -7, "fused decode attention")
    assert torch.equal(cache[:, :, :, t], ref_cache[:, :, :, t]), "appended key / value row"
    assert torch.equal(cache[:, :, :, :t], ref_cache[:, :, :, :t]) and torch.equal(cache[:, :, :, t + 1:], ref_cache[:, :, :, t + 1:])


@pytest.mark.parametrize("tile_m", [256, 193])
def test_gemm_pipelined_gather(dev, tile_m):
    """The gathered-A instance of the persistent pipelined kernel (scalar index loads): 27-tap Conv3d rows with -1 = zero
    row, and a one-tap row gather with M not a multiple of the tile, against the two-barrier kernel on the same operands."""
    from grove_amd import _lib, ops
    from grove_amd.model.indexing import conv3d_gather_index
    L = _lib.lib()
    G, T, H, W, Ci, Co = 2, 8, 8, 9, 64, 264
    M = G * T * H * W
    x, w, bias = rnd(M, Ci, seed=11).to(dev), rnd(Co, 27 * Ci, seed=12, scale=0.05).to(dev), rnd(Co, seed=13).to(dev)
    res = rnd(M, Co, seed=14).to(dev)
    idx = conv3d_gather_index(G, T, H, W).to(dev)
    kw = dict(act=ops.ACT_RELU, residual=res, scale_ptr=torch.tensor([0.3]).to(dev), scale_tanh=True, a_idx=idx, a_taps=27, M=M)
    try:
        L.grove_gemm_set_tile_m(128)
        ref = ops.linear(x, w, bias, **kw)
        assert L.grove_gemm_last_variant() in (1, 2, 3)
        L.grove_gemm_set_tile_m(tile_m)
        out = ops.linear(x, w, bias, **kw)
        assert L.grove_gemm_last_variant() == (6 if tile_m == 256 else 7)
        assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
        # one tap, rows permuted / dropped (window partition), M = 1000 (edge tile), 8-row groups crossing the array end
        M2, K2, N2 = 1000, 128, 520
        a, b2 = rnd(1300, K2, seed=15).to(dev), rnd(N2, K2, seed=16, scale=0.05).to(dev)
        g = torch.Generator().manual_seed(3)
        idx2 = torch.randint(0, 1300, (M2,), generator=g, dtype=torch.int32)
        idx2[::7] = -1
        idx2 = idx2.to(dev)
        L.grove_gemm_set_tile_m(128)
        ref2 = ops.linear(a, b2, a_idx=idx2, a_taps=1, M=M2)
        L.grove_gemm_set_tile_m(tile_m)
        out2 = ops.linear(a, b2, a_idx=idx2, a_taps=1, M=M2)
        assert L.grove_gemm_last_variant() == (6 if tile_m == 256 else 7)
        assert torch.equal(out2, ref2)
        gathered = torch.where((idx2 >= 0)[:, None], a.float()[idx2.clamp_min(0).long()], torch.zeros(1, device=dev))
        close(out2, gathered.cpu() @ b2.float().cpu().t(), 2 ** -7, "one-tap gather vs fp32")
    finally:
        L.grove_gemm_set_tile_m(0)


def test_wgrad_tn_pipelined(dev):
    """The persistent pipelined TN kernel (K-major LDS-DMA images, swizzled transposed reads) against the 128 x 128 kernel and
    fp32: plain operands with edge tiles in M and N, and the gathered Conv3d weight gradient (27 taps, zero rows)."""
    from grove_amd import _lib, ops
    from grove_amd.model.indexing import conv3d_gather_index
    L = _lib.lib()
    try:
        K, M, N = 1216, 328, 520
        dy, x = rnd(K, M, seed=90).to(dev), rnd(K, N, seed=91).to(dev)
        g0 = torch.randn(M, N, generator=torch.Generator().manual_seed(92))
        L.grove_gemm_tn_set_pipelined(0)
        ref = ops.wgrad(dy, x, g0.clone().to(dev), alpha=0.5)
        L.grove_gemm_tn_set_pipelined(1)
        out = ops.wgrad(dy, x, g0.clone().to(dev), alpha=0.5)
        close(out, g0 + 0.5 * dy.float().cpu().t() @ x.float().cpu(), 3e-5, "pipelined wgrad vs fp32")
        close(out, ref, 2e-6, "pipelined wgrad vs 128x128 kernel")
        # Conv3d weight gradient: Ci = 256 so that a 256-column tile stays inside one tap
        G, T, H, W, Ci, Co = 2, 4, 4, 8, 256, 264
        Mtok = G * T * H * W
        xx, dz = rnd(Mtok, Ci, seed=93).to(dev), rnd(Mtok, Co, seed=94).to(dev)
        idx = conv3d_gather_index(G, T, H, W).to(dev)
        L.grove_gemm_tn_set_pipelined(0)
        ref = ops.wgrad(dz, xx, torch.zeros(Co, 27 * Ci, dtype=torch.float32, device=dev), b_idx=idx, b_taps=27)
        L.grove_gemm_tn_set_pipelined(1)
        out = ops.wgrad(dz, xx, torch.zeros(Co, 27 * Ci, dtype=torch.float32, device=dev), b_idx=idx, b_taps=27)
        close(out, ref, 2e-6, "pipelined conv3d wgrad vs 128x128 kernel")
        # a partial last round cut into K ranges (fp32 atomics): 300 tiles on 256 CUs, plain and gathered
        Kt, M2, N2 = 8192, 1280, 15360
        dy, x = rnd(Kt, M2, seed=95).to(dev), rnd(Kt, N2, seed=96).to(dev)
        L.grove_gemm_tn_set_split_tail(0)
        ref = ops.wgrad(dy, x, torch.ones(M2, N2, dtype=torch.float32, device=dev), alpha=0.5)
        assert L.grove_gemm_tn_last_parts() == 1
        L.grove_gemm_tn_set_split_tail(2)
        out = ops.wgrad(dy, x, torch.ones(M2, N2, dtype=torch.float32, device=dev), alpha=0.5)
        assert L.grove_gemm_tn_last_parts() > 1, "the tail of 44 tiles is cut"
        close(out, ref, 2e-6, "cut last round vs whole tiles")
        G, T, H, W, Ci, Co = 2, 4, 32, 32, 512, 1280
        xx, dz = rnd(G * T * H * W, Ci, seed=97).to(dev), rnd(G * T * H * W, Co, seed=98).to(dev)
        idx = conv3d_gather_index(G, T, H, W).to(dev)
        L.grove_gemm_tn_set_split_tail(0)
        ref = ops.wgrad(dz, xx, torch.zeros(Co, 27 * Ci, dtype=torch.float32, device=dev), b_idx=idx, b_taps=27)
        L.grove_gemm_tn_set_split_tail(1)
        out = ops.wgrad(dz, xx, torch.zeros(Co, 27 * Ci, dtype=torch.float32, device=dev), b_idx=idx, b_taps=27)
        assert L.grove_gemm_tn_last_parts() > 1
        close(out, ref, 2e-6, "cut last round, gathered taps")
    finally:
        L.grove_gemm_tn_set_pipelined(-1)
        L.grove_gemm_tn_set_split_tail(1)


@pytest.mark.parametrize("M,I,K,save", [(2812, 1024, 256, True), (1500, 2752, 128, False)])
def test_gemm_swiglu_pair_epilogue(dev, M, I, K, save):
    """LlamaMLP's silu(gate) * up folded into the gate|up GEMM (interleaved weight rows, ACT_SWIGLU_PAIR): bit-identical to the
    GEMM + grove_swiglu_fwd pair, and the optional aux output is the un-interleaved gate | up activation."""
    from grove_amd import ops
    x, wgu = rnd(M, K, seed=1).to(dev), rnd(2 * I, K, seed=2, scale=0.2).to(dev)
    gu_ref = ops.linear(x, wgu)
    a_ref = ops.swiglu(gu_ref, I)
    gu = torch.empty((M, 2 * I), dtype=bf16, device=dev) if save else None
    a = ops.linear(x, ops.swiglu_interleave(wgu), act=ops.ACT_SWIGLU_PAIR, aux=gu, ld_aux=2 * I)
    assert a.shape == (M, I)
    assert torch.equal(a, a_ref), (a.float() - a_ref.float()).abs().max().item()
    if save:
        assert torch.equal(gu, gu_ref)
    xf, wf = x.float().cpu(), wgu.float().cpu()
    g, u = xf @ wf[:I].t(), xf @ wf[I:].t()
    close(a, torch.nn.functional.silu(g) * u, 2 ** -6, "swiglu pair vs fp32")


def test_gemm_padded_head_maps(dev):
    """n_group/n_pad (compact N written into a padded-head layout, pad columns zeroed) and k_group/k_pad (padded-head A read as
    compact K) of the pipelined kernel, against the same GEMMs on zero-padded weights (SAM: head dim 80 stored as 96)."""
    from grove_amd import ops
    heads, hd, hp, C, M, Mw = 8, 80, 96, 256, 3000, 4100  # 8 x 80 = 640: the compact K must stay a multiple of 64
    g = torch.Generator().manual_seed(5)
    tok2win = torch.randperm(Mw, generator=g)[:M].to(torch.int32).to(dev)
    x = rnd(M, C, seed=1).to(dev)
    w_c = rnd(heads * hd, C, seed=2, scale=0.1).to(dev)          # compact [480, C]
    b_c = rnd(heads * hd, seed=3).to(dev)
    w_p = torch.zeros(heads, hp, C, dtype=bf16, device=dev)
    w_p[:, :hd] = w_c.view(heads, hd, C)
    b_p = torch.zeros(heads, hp, dtype=bf16, device=dev)
    b_p[:, :hd] = b_c.view(heads, hd)
    # N map + row scatter: compact weights, padded output with garbage pre-filled (pad columns must come out zero)
    ref = ops.linear(x, w_p.view(heads * hp, C), b_p.view(-1), c_idx=tok2win, out=torch.zeros(Mw, heads * hp, dtype=bf16, device=dev))
    out = torch.full((Mw, heads * hp), 7.0, dtype=bf16, device=dev)
    ops.linear(x, w_c, b_c, c_idx=tok2win, out=out, n_map=(hd, hp - hd))
    rows = tok2win.long()
    assert torch.equal(out[rows], ref[rows])
    # K map + row gather: padded-head A, compact K weights
    a_p = torch.zeros(Mw, heads, hp, dtype=bf16, device=dev)
    a_p[:, :, :hd] = rnd(Mw, heads, hd, seed=4).to(dev)
    a_p[:, :, hd:] = 3.0                                         # pad columns are never read
    w2_c = rnd(C, heads * hd, seed=6, scale=0.1).to(dev)
    w2_p = torch.zeros(C, heads, hp, dtype=bf16, device=dev)
    w2_p[:, :, :hd] = w2_c.view(C, heads, hd)
    a_ref = a_p.clone()
    a_ref[:, :, hd:] = 0
    ref2 = ops.linear(a_ref.view(Mw, heads * hp), w2_p.view(C, heads * hp), a_idx=tok2win, a_taps=1, M=M)
    out2 = ops.linear(a_p.view(Mw, heads * hp), w2_c, a_idx=tok2win, a_taps=1, M=M, k_map=(hd, hp - hd))
    close(out2, ref2, 2 ** -8, "k map vs padded GEMM")  # (K = 640 vs 768: same products, the zero columns dropped)
    want = a_ref.view(Mw, heads * hp)[rows].float().cpu() @ w2_p.view(C, heads * hp).float().cpu().t()
    close(out2, want, 2 ** -7, "k map vs fp32")


def test_adamw_multi_tensor_equals_per_tensor(dev):
    """The one-launch optimizer step over a flat state buffer with separate bf16 tensors (ragged sizes, 16-byte aligned
    starts with gaps, one tensor without a working copy) is bit-identical to per-tensor grove_adamw_step calls."""
    from grove_amd import ops
    sizes = [5, 1024, 3000, 1, 77777, 4096]
    offs, off = [], 0
    for k in sizes:
        offs.append(off)
        off += (k + 3) // 4 * 4
    total = off
    g = torch.Generator().manual_seed(21)
    state = {n: torch.zeros(total) for n in ("master", "grad", "m", "v")}
    for o, k in zip(offs, sizes):
        state["master"][o:o + k] = torch.randn(k, generator=g)
        state["grad"][o:o + k] = torch.randn(k, generator=g) * 0.1
        state["m"][o:o + k] = torch.randn(k, generator=g) * 0.01
        state["v"][o:o + k] = torch.rand(k, generator=g) * 0.01
    hp = dict(lr=3e-4, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.01, grad_scale=0.37, step=7)
    a = {n: t.clone().to(dev) for n, t in state.items()}
    b = {n: t.clone().to(dev) for n, t in state.items()}
    wa = [torch.zeros(k, dtype=bf16, device=dev) for k in sizes]
    wb = [torch.zeros(k, dtype=bf16, device=dev) for k in sizes]
    for i, (o, k) in enumerate(zip(offs, sizes)):
        ops.adamw_step(a["master"][o:o + k], wa[i] if i != 3 else None, a["grad"][o:o + k], a["m"][o:o + k], a["v"][o:o + k], **hp)
    seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    seg_len = torch.tensor(sizes, dtype=torch.int64, device=dev)
    ptrs = torch.tensor([w.data_ptr() if i != 3 else 0 for i, w in enumerate(wb)], dtype=torch.int64, device=dev)
    ops.adamw_step_multi(b["master"], b["grad"], b["m"], b["v"], seg_off, seg_len, ptrs, **hp)
    for n in ("master", "m", "v"):
        assert torch.equal(a[n], b[n]), n
    for i in range(len(sizes)):
        assert torch.equal(wa[i], wb[i]), i
    assert (wb[3] == 0).all()


# ----------------------------------------------------------------------------- fp8 path (config 5)
def _e4m3(x):
    """torch's OCP e4m3fn rounding of an fp32 tensor, back in fp32 (the CPU model of grove_quant_fp8_rows's cast)."""
    return x.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def test_fp8_gemm_operand_map_exact_integers(dev):
    """Small integers are exact in e4m3, unit scales: the MFMA operand map of gemm_fp8.hip (row = lane & 15, k = 32 (lane >> 4) + j),
    the XOR-swizzled LDS image and the tile order must reproduce the integer product EXACTLY; B is asymmetric."""
    from grove_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 200, 264, 384
    a = torch.randint(-2, 3, (M, K), generator=g).float() * (torch.rand(M, K, generator=g) < 0.2).float()
    b = torch.randint(-3, 4, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.3).float()
    b[:, 0] += (torch.arange(N) % 5 == 0).float()
    aq = a.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    bq = b.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    one_m, one_n = torch.ones(M, device=dev), torch.ones(N, device=dev)
    ref = a @ b.t()
    assert ref.abs().max() < 256  # exact in bf16
    from grove_amd import _lib
    try:
        for pipelined in (1, 0):  # the FP8 instance of the persistent pipelined kernel (gemm.hip), then gemm_fp8.hip's own kernel
            _lib.lib().grove_gemm_fp8_set_pipelined(pipelined)
            y = ops.linear_fp8(None, bq, one_n, xq=(aq, one_m))
            assert torch.equal(y.float().cpu(), ref), (pipelined, (y.float().cpu() - ref).abs().max())
    finally:
        _lib.lib().grove_gemm_fp8_set_pipelined(1)


@pytest.mark.parametrize("M,N,K,act", [(333, 520, 1024, 0), (2812, 4096, 4096, 0), (1000, 1024, 512, 1), (1000, 1024, 512, 2), (20200, 1000, 5120, 2),
                                       (20200, 1000, 5120, 0)])
def test_fp8_linear_quantised(dev, M, N, K, act):
    """quant_fp8_rows (per-row amax / 448, e4m3) + the fp8 GEMM with bias / activation / residual, against the same quantisation
    modelled with torch's float8_e4m3fn casts (tight), and against the unquantised product (the fp8 error itself: a few percent)."""
    from grove_amd import ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    xq, xs = ops.quant_fp8_rows(x.to(dev))
    wq, ws = ops.quant_fp8_rows(w.to(dev))
    xs_ref = x.float().abs().amax(1) / 448
    assert torch.allclose(xs.cpu(), xs_ref, rtol=1e-6)
    x_model = _e4m3(x.float() / xs_ref[:, None])
    same = (xq.cpu().view(torch.float8_e4m3fn).float() == x_model).float().mean().item()
    assert same > 0.999, same  # (ties of the device's division vs torch's can differ in the last bit on a handful of elements)
    ws_ref = w.float().abs().amax(1) / 448
    w_model = _e4m3(w.float() / ws_ref[:, None])
    a = {0: ops.ACT_NONE, 1: ops.ACT_GELU, 2: ops.ACT_QUICKGELU}[act]  # (round 6: GELU has its pipelined FP8 instances too — SAM's mlp.lin1)
    fact = {0: lambda v: v, 1: torch.nn.functional.gelu, 2: lambda v: v * torch.sigmoid(1.702 * v)}[act]
    y = ops.linear_fp8(x.to(dev), wq, ws, bias.to(dev), act=a, residual=res.to(dev))
    pre = (x_model @ w_model.t()) * xs_ref[:, None] * ws_ref[None, :] + bias.float()
    ref = fact(pre) + res.float()
    rms = ((y.float().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rms < 6e-3, f"fp8 gemm vs its own quantisation model: rms {rms}"  # bf16 output rounding + last-bit ties of the device division (measured 3.0-3.4e-3)
    close(y, ref, 2e-2, "fp8 gemm vs its own quantisation model")
    full = fact(x.float() @ w.float().t() + bias.float()) + res.float()
    if True:  # both kernels: same products, different fp32 sum order
        from grove_amd import _lib
        try:
            _lib.lib().grove_gemm_fp8_set_pipelined(0)
            y0 = ops.linear_fp8(x.to(dev), wq, ws, bias.to(dev), act=a, residual=res.to(dev))
        finally:
            _lib.lib().grove_gemm_fp8_set_pipelined(1)
        close(y, y0, 2 ** -7, "pipelined FP8 instance vs the two-barrier fp8 kernel")
    err = ((y.float().cpu() - full).pow(2).mean().sqrt() / full.pow(2).mean().sqrt()).item()
    assert err < 6e-2, f"fp8 quantisation error {err}"


@pytest.mark.parametrize("rows,K,ld", [(37, 5120, 5120), (9, 520, 528), (5, 8192, 8192), (130, 1280, 1280)])
def test_fp8_quantisation_with_gelu_in_front(dev, rows, K, ld):
    """grove_quant_fp8_rows_act: e4m3(gelu(x) / scale) with the per-row scale of gelu(x) — equal to quantising the stored bf16 GELU output
    (what the unfused path does), up to last-bit differences of the erf approximation at bf16 rounding boundaries."""
    from grove_amd import ops
    x = rnd(rows, ld, seed=7, scale=2.0)[:, :K]
    xd = x.to(dev)
    q, sc = ops.quant_fp8_rows(xd, act=ops.ACT_GELU)
    f16 = torch.nn.functional.gelu(x.float()).to(bf16)
    q0, sc0 = ops.quant_fp8_rows(f16.to(dev).contiguous())
    assert torch.allclose(sc.cpu(), sc0.cpu(), rtol=2 ** -7)
    a, b = q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None], q0.cpu().view(torch.float8_e4m3fn).float() * sc0.cpu()[:, None]
    assert (q.cpu() == q0.cpu()).float().mean().item() > 0.995
    close(a, b, 2 ** -3 * 1.01, "gelu + quantise vs quantise of the stored gelu")  # (a flipped bf16 bit can move a code by one e4m3 step)
    close(a, torch.nn.functional.gelu(x.float()), 2 ** -4 * 1.1, "e4m3 rounding of gelu(x)")
    with pytest.raises(RuntimeError):
        ops.quant_fp8_rows(rnd(2, 8704, seed=1).to(dev), act=ops.ACT_GELU)  # K > 8192: the row does not fit the registers


@pytest.mark.parametrize("nb,nh,size,hd,hp", [(18, 16, 14, 80, 96), (3, 16, 32, 80, 96), (5, 4, 6, 32, 32), (7, 12, 14, 64, 64)])
def test_rel_bias_streams(dev, nb, nh, size, hd, hp):
    """The matrix-core rel-pos streams (grove_rel_bias_fwd / _bwd) against fp32 and against the GEMM batched over the query
    positions they replace: SAM-H window (14 x 14, hd 80 padded to 96, 32 bins) and global (32 x 32, 64 bins) shapes, fewer
    than 16 heads, window counts that are not a multiple of the per-wave chunk. The backward accumulates into dq in place and
    must leave the pad dims hd..hp-1 and the k | v columns of the fused buffer untouched."""
    from grove_amd import ops
    from grove_amd.model.sam import _rcat_tables
    L = size * size
    g = torch.Generator().manual_seed(nb * 1000 + size)
    rel_h = (torch.randn(2 * size - 1, hd, generator=g) * 0.5).to(bf16).to(dev)
    rel_w = (torch.randn(2 * size - 1, hd, generator=g) * 0.5).to(bf16).to(dev)
    rcat, rcat_t, khp, rel_ld = _rcat_tables(size, rel_h, rel_w, hd, hp, hd ** -0.5)
    assert ops.rel_bias_applicable(nh, hp, rel_ld)
    ld = 3 * nh * hp
    qkv = (torch.randn(nb * L, ld, generator=g) * 0.5).to(bf16).to(dev)
    qkv.view(nb * L, 3 * nh, hp)[:, :, hd:] = 0  # pad dims: zero, as the qkv GEMM writes them
    hrow = (torch.arange(nb, dtype=torch.int32)[:, None] * (L * (ld // hp)) + torch.arange(nh, dtype=torch.int32)[None, :]).reshape(-1).to(dev)
    # forward
    rel = ops.rel_bias_fwd(qkv, rcat, nb, nh, L, hp, hd)
    ref_g = torch.empty((nb * nh, L, rel_ld), dtype=bf16, device=dev)
    ops.gemm_raw(qkv, rcat, ref_g, nb * nh, rel_ld, hp, hp, hp, L * rel_ld, a_idx=hrow, batch=(L, 1), sA=(ld, 0), sB=(rel_ld * hp, 0), sC=(rel_ld, 0))
    q4 = qkv.view(nb, L, 3 * nh, hp)[:, :, :nh].float()                      # [nb, L, nh, hp]
    ref = torch.einsum("bqhd,qnd->bhqn", q4, rcat.float()).reshape(nb * nh, L, rel_ld)
    close(rel, ref, 2 ** -7, "rel' vs fp32")
    assert (rel.float() - ref_g.float()).abs().max().item() <= 2 ** -7 * ref.abs().max().item(), "rel' vs the batched GEMM"
    # backward
    drel = (torch.randn(nb * nh, L, rel_ld, generator=g) * 0.3).to(bf16).to(dev)
    dq0 = (torch.randn(nb * L, ld, generator=g) * 0.5).to(bf16).to(dev)
    dq = dq0.clone()
    ops.rel_bias_bwd(drel, rcat_t, dq, nb, nh, L, hp, hd)
    dq_g = dq0.clone()
    ops.gemm_raw(drel, rcat_t, dq_g, nb * nh, hp, rel_ld, L * rel_ld, rel_ld, hp, c_idx=hrow, residual=dq_g, ldr=hp,
                 batch=(L, 1), sA=(rel_ld, 0), sB=(hp * rel_ld, 0), sC=(ld, 0), sR=(ld, 0))
    add = torch.einsum("bhqn,qdn->bqhd", drel.float().view(nb, nh, L, rel_ld), rcat_t.float())  # [nb, L, nh, hp]
    want = dq0.float().view(nb, L, 3 * nh, hp).clone()
    want[:, :, :nh] += add
    close(dq, want.view(nb * L, ld), 2 ** -7, "dq += d rel' . Rcat vs fp32")
    assert (dq.float() - dq_g.float()).abs().max().item() <= 2 ** -7 * want.abs().max().item(), "dq vs the batched GEMM"
    untouched = torch.ones(3 * nh, hp, dtype=torch.bool)
    untouched[:nh, :hd] = False
    assert torch.equal(dq.view(nb * L, 3 * nh, hp)[:, untouched.to(dev)], dq0.view(nb * L, 3 * nh, hp)[:, untouched.to(dev)]), "columns outside q[:hd]"


@pytest.mark.parametrize("valid", [[(14, 14), (4, 14), (14, 4), (4, 4)], [(1, 1), (3, 7), (14, 1), (9, 14)]])
def test_window_attention_valid_queries(dev, valid):
    """q_valid of the window kernels: window b keeps its top-left vy x vx positions as queries (window_partition's padding is real as
    keys, dropped as queries). Against fp32 attention over ALL keys at the valid queries: o / lse forward; dq / d rel' at the valid
    queries and dk / dv everywhere backward, with NaN in d_o at the padded positions (never read) and NaN-filled outputs (rows at
    padded positions stay untouched). The window shapes of a 32 x 32 grid — 14 x 14 (full), 4 x 14, 14 x 4, 4 x 4 — and odd ones (a single
    query, one column, partial tiles in both query loops)."""
    from grove_amd import ops
    B, H, L, hs, hd, ws = 4, 16, 196, 96, 80, 14
    assert ops.window_kernels_take(L, hs, hd, 32)
    g = torch.Generator().manual_seed(77)
    alpha = hd ** -0.5
    qkv = torch.zeros(B * L, 3, H, hs)
    qkv[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g) * 0.7
    qkv = qkv.view(B * L, 3 * H * hs).to(bf16)
    relp = torch.zeros(B * H, L, 32)
    relp[..., :ws] = torch.randn(B * H, L, ws, generator=g) / alpha
    relp[..., 16:16 + ws] = torch.randn(B * H, L, ws, generator=g) / alpha
    relp = relp.to(bf16)
    do = torch.zeros(B * L, H, hs)
    do[..., :hd] = torch.randn(B * L, H, hd, generator=g) * 0.5  # (pad dims: zero, as the projection's dgrad writes them)
    do = do.view(B * L, H * hs).to(bf16)
    qmask = torch.zeros(B, ws, ws, dtype=torch.bool)
    for b, (vy, vx) in enumerate(valid):
        qmask[b, :vy, :vx] = True
    qmask = qmask.view(B, L)
    # fp32 reference: every key, gradients only from the valid queries
    t = qkv.float().view(B, L, 3, H, hs).requires_grad_(True)
    q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)
    relr = relp.float().clone().requires_grad_(True)
    s = q @ k.transpose(-1, -2) * alpha + ((relr[..., :ws, None] + relr[..., None, 16:16 + ws]) * alpha).reshape(B, H, L, L)
    o_ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H * hs)
    do_ref = do.float() * qmask.view(B * L, 1)
    o_ref.backward(do_ref)
    lse_ref = torch.logsumexp(s, -1).reshape(B * H, L)
    qv = torch.tensor(valid, dtype=torch.int32).to(dev)
    out = torch.full((B * L, H * hs), float("nan"), dtype=bf16, device=dev)
    out, lse = ops.flash_attn(qkv.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(16, ws), want_lse=True,
                              hs_valid=hd, q_valid=qv, out=out)
    rows = qmask.view(-1)
    close(out[rows.to(dev)], o_ref[rows], 1e-2, "o at the valid queries")
    assert torch.isnan(out[(~rows).to(dev)].float()).all(), "rows of o at padded positions must not be written"
    hm = qmask[:, None, :].expand(B, H, L).reshape(B * H, L)
    close(lse[hm.to(dev)], lse_ref[hm], 2e-3, "lse at the valid queries")
    do_dev = do.clone()
    do_dev[~rows] = float("nan")
    dqkv = torch.full((B * L, 3 * H * hs), float("nan"), dtype=bf16, device=dev)
    drel = ops.flash_attn_bwd(qkv.to(dev), out, do_dev.to(dev), lse, dqkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev),
                              rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv)
    gref = t.grad.reshape(B * L, 3 * H * hs)
    close(dqkv[rows.to(dev), :H * hs], gref[rows, :H * hs], 2e-2, "dq at the valid queries")
    close(dqkv[:, H * hs:2 * H * hs], gref[:, H * hs:2 * H * hs], 2e-2, "dk")
    close(dqkv[:, 2 * H * hs:], gref[:, 2 * H * hs:], 2e-2, "dv")
    close(drel[hm.to(dev)], relr.grad[hm], 2e-2, "d rel' at the valid queries")
    assert torch.isnan(dqkv[(~rows).to(dev), :H * hs].float()).all(), "rows of dq at padded positions must not be written"
    # pad_row: k / v of the padded positions come from ONE row (the bias); the rows of qkv there are not read (NaN) and dk / dv there
    # are not written. Reference = the same run with the row copied into every padded position.
    pad_row = torch.zeros(3, H, hs)
    pad_row[..., :hd] = torch.randn(3, H, hd, generator=g) * 0.7
    pad_row = pad_row.view(-1).to(bf16)
    filled = qkv.clone()
    filled[~rows] = pad_row
    o_f, lse_f = ops.flash_attn(filled.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(16, ws), want_lse=True,
                                hs_valid=hd, q_valid=qv)
    dq_f = torch.full((B * L, 3 * H * hs), float("nan"), dtype=bf16, device=dev)
    dr_f = ops.flash_attn_bwd(filled.to(dev), o_f, do_dev.to(dev), lse_f, dq_f, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev),
                              rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv)
    holes = qkv.clone()
    holes[~rows] = float("nan")
    o_p, lse_p = ops.flash_attn(holes.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(16, ws), want_lse=True,
                                hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev))
    dq_p = torch.full((B * L, 3 * H * hs), float("nan"), dtype=bf16, device=dev)
    dr_p = ops.flash_attn_bwd(holes.to(dev), o_p, do_dev.to(dev), lse_p, dq_p, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev),
                              rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev))
    rd = rows.to(dev)
    assert torch.equal(o_p[rd], o_f[rd]) and torch.equal(lse_p[hm.to(dev)], lse_f[hm.to(dev)]), "forward with the padded keys taken from pad_row"
    assert torch.equal(dq_p[rd], dq_f[rd]) and torch.equal(dr_p[hm.to(dev)], dr_f[hm.to(dev)]), "backward with the padded keys taken from pad_row"
    assert torch.isnan(dq_p[~rd].float()).all(), "dq / dk / dv rows at padded positions must not be written"
    # o_map: o and d_o in TOKEN order with compact heads (row o_map[b * L + i], head h at column h * hd) — window_unpartition folded
    # into the kernels' addressing. Same numbers as the windowed-layout run, at permuted rows.
    ntok = int(rows.sum())
    tok_of = torch.full((B * L,), -1, dtype=torch.int32)
    tok_of[rows] = torch.randperm(ntok, generator=g).to(torch.int32)
    omap = tok_of.to(dev)
    o_t = torch.full((ntok, H * hd), float("nan"), dtype=bf16, device=dev)
    o_t, lse_t = ops.flash_attn(holes.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev), rel_hw=(16, ws), want_lse=True,
                                hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev), o_map=omap, o_rows=ntok, out=o_t)
    sel = tok_of[rows].long().to(dev)
    assert torch.equal(o_t[sel].view(ntok, H, hd), o_p[rd].view(ntok, H, hs)[..., :hd]), "token-order o"
    assert torch.equal(lse_t[hm.to(dev)], lse_p[hm.to(dev)])
    do_t = torch.zeros(ntok, H * hd, dtype=bf16, device=dev)
    do_t[sel] = do_dev.to(dev)[rd].view(ntok, H, hs)[..., :hd].reshape(ntok, H * hd)
    dq_t = torch.full((B * L, 3 * H * hs), float("nan"), dtype=bf16, device=dev)
    dr_t = ops.flash_attn_bwd(holes.to(dev), o_t, do_t, lse_t, dq_t, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev),
                              rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev), o_map=omap)
    assert torch.equal(dq_t[rd], dq_p[rd]) and torch.equal(dr_t[hm.to(dev)], dr_p[hm.to(dev)]), "backward from token-order o / d_o"
    # g_tok (round 6): dq / dk / dv in token order with compact heads too — [tokens, q | k | v blocks of H * hd columns] — the same numbers
    # as the windowed-layout run at the mapped rows, every element of the tensor written (a torch.empty buffer is enough)
    dg = torch.full((ntok, 3 * H * hd), float("nan"), dtype=bf16, device=dev)
    dr_g = ops.flash_attn_bwd(holes.to(dev), o_t, do_t, lse_t, dg, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=relp.to(dev),
                              rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev), o_map=omap, grads_tok=True)
    assert torch.equal(dr_g[hm.to(dev)], dr_p[hm.to(dev)])
    assert not torch.isnan(dg.float()).any(), "every element of the token-order gradient must be written"
    want = dq_p[rd].view(ntok, 3, H, hs)[..., :hd]
    assert torch.equal(dg[sel].view(ntok, 3, H, hd), want), "token-order dq / dk / dv"


@pytest.mark.parametrize("valid", [[(14, 14), (4, 14), (14, 4), (4, 4)], [(1, 1), (3, 7), (14, 1), (9, 14)]])
def test_window_attention_rel_pos_in_kernel(dev, valid):
    """grove_flash_attn_params.rel_table (round 6b): the decomposed rel-pos terms made INSIDE the window kernels from the 27 + 27
    embeddings (image_encoder.py:387-458) — against fp32 attention with add_decomposed_rel_pos written out, and against the two-stream
    form (grove_rel_bias_fwd -> kernel -> grove_rel_bias_bwd) on the same inputs: o, lse, the operand tensor the forward leaves, dq
    (with its rel-pos term), dk, dv; padded queries skipped, k / v of padded positions from pad_row, windowed and token-order layouts."""
    from grove_amd import ops
    from grove_amd.model.sam import _rcat_tables
    B, H, L, hs, hd, ws = 4, 16, 196, 96, 80, 14
    g = torch.Generator().manual_seed(123)
    alpha = hd ** -0.5
    rel_h = (torch.randn(2 * ws - 1, hd, generator=g) * 0.2).to(bf16)
    rel_w = (torch.randn(2 * ws - 1, hd, generator=g) * 0.2).to(bf16)
    T = ops.rel_table_images(rel_h.to(dev), rel_w.to(dev), ws, alpha)
    rcat, rcat_t, khp, rel_ld = _rcat_tables(ws, rel_h.to(dev), rel_w.to(dev), hd, hs, alpha)
    assert khp == 16 and rel_ld == 32
    qkv = torch.zeros(B * L, 3, H, hs)
    qkv[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g) * 0.7
    qkv = qkv.view(B * L, 3 * H * hs).to(bf16)
    do = torch.zeros(B * L, H, hs)
    do[..., :hd] = torch.randn(B * L, H, hd, generator=g) * 0.5
    do = do.view(B * L, H * hs).to(bf16)
    qmask = torch.zeros(B, ws, ws, dtype=torch.bool)
    for b, (vy, vx) in enumerate(valid):
        qmask[b, :vy, :vx] = True
    qmask = qmask.view(B, L)
    rows = qmask.view(-1)
    pad_row = torch.zeros(3, H, hs)
    pad_row[..., :hd] = torch.randn(3, H, hd, generator=g) * 0.7
    pad_row = pad_row.view(-1).to(bf16)
    filled = qkv.clone()
    filled[~rows] = pad_row
    # fp32 reference: add_decomposed_rel_pos written out
    t = filled.float().view(B, L, 3, H, hs).requires_grad_(True)
    q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)   # [B, H, L, hs]
    c = torch.arange(ws)
    Rh = rel_h.float()[(c[:, None] - c[None, :]) + (ws - 1)]   # [qy, ky, hd]
    Rw = rel_w.float()[(c[:, None] - c[None, :]) + (ws - 1)]
    r_q = q[..., :hd].reshape(B, H, ws, ws, hd)
    relh = torch.einsum("bnhwc,hkc->bnhwk", r_q, Rh)
    relw = torch.einsum("bnhwc,wkc->bnhwk", r_q, Rw)
    s = (q @ k.transpose(-1, -2) * alpha).view(B, H, ws, ws, ws, ws) + relh[..., :, None] + relw[..., None, :]
    s = s.view(B, H, L, L)
    o_ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H * hs)
    o_ref.backward(do.float() * rows.view(-1, 1))
    lse_ref = torch.logsumexp(s, -1).reshape(B * H, L)
    gref = t.grad.reshape(B * L, 3 * H * hs)
    qv = torch.tensor(valid, dtype=torch.int32).to(dev)
    hm = qmask[:, None, :].expand(B, H, L).reshape(B * H, L)
    holes = qkv.clone()
    holes[~rows] = float("nan")
    do_dev = do.clone()
    do_dev[~rows] = float("nan")
    # the two-stream form
    rel_s = ops.rel_bias_fwd(holes.to(dev), rcat, B, H, L, hs, hd, q_valid=qv, kw=ws)
    o_s, lse_s = ops.flash_attn(holes.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel_s, rel_hw=(16, ws), want_lse=True,
                                hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev))
    dq_s = torch.full((B * L, 3 * H * hs), float("nan"), dtype=bf16, device=dev)
    dr_s = ops.flash_attn_bwd(holes.to(dev), o_s, do_dev.to(dev), lse_s, dq_s, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel_s,
                              rel_hw=(16, ws), want_drel=True, hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev))
    ops.rel_bias_bwd(dr_s, rcat_t, dq_s, B, H, L, hs, hd, q_valid=qv, kw=ws)
    # in the kernels
    rel_o = torch.full((B * H, L, 32), float("nan"), dtype=bf16, device=dev)
    o_k = torch.full((B * L, H * hs), float("nan"), dtype=bf16, device=dev)
    o_k, lse_k = ops.flash_attn(holes.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel_hw=(16, ws), want_lse=True, hs_valid=hd, q_valid=qv,
                                pad_row=pad_row.to(dev), rel_table=T, rel_out=rel_o, out=o_k)
    rd = rows.to(dev)

    def err(a, b_):
        return (a.float().cpu() - b_.float().cpu()).abs().max().item()
    # both forms round the bias operand to bf16 once more than the fp32 reference: the in-kernel form must be as close to it as the streams
    e_k, e_s = err(o_k[rd], o_ref[rows]), err(o_s[rd], o_ref[rows])
    assert e_k <= max(1.5 * e_s, 1e-2 * o_ref.abs().max().item()), f"o vs fp32: in kernel {e_k:.4g}, two streams {e_s:.4g}"
    close(o_k[rd], o_s[rd], 2e-2, "o vs the two-stream form")
    assert torch.isnan(o_k[~rd].float()).all(), "rows of o at padded positions must not be written"
    close(lse_k[hm.to(dev)], lse_ref[hm], 2e-3, "lse vs fp32")
    sc = alpha * 1.4426950408889634
    close(rel_o[hm.to(dev)], rel_s[hm.to(dev)].float() * sc, 2e-2, "the operand tensor = rel' x alpha log2 e")
    assert torch.isnan(rel_o[(~hm).to(dev)].float()).all(), "operand rows at padded positions must not be written"
    no_table = ops.flash_attn(holes.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel_hw=(16, ws), hs_valid=hd, q_valid=qv,
                              pad_row=pad_row.to(dev), rel_table=T)[0]   # inference: nothing kept
    assert torch.equal(no_table[rd], o_k[rd])
    dq_k = torch.full((B * L, 3 * H * hs), float("nan"), dtype=bf16, device=dev)
    ops.flash_attn_bwd(holes.to(dev), o_k, do_dev.to(dev), lse_k, dq_k, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel_o, rel_hw=(16, ws),
                       hs_valid=hd, q_valid=qv, pad_row=pad_row.to(dev), rel_table=T)
    for name, lo in (("dq", 0), ("dk", H * hs), ("dv", 2 * H * hs)):
        want = gref[rows, lo:lo + H * hs]
        e_k, e_s = err(dq_k[rd, lo:lo + H * hs], want), err(dq_s[rd, lo:lo + H * hs], want)
        assert e_k <= max(1.5 * e_s, 2e-2 * want.abs().max().item()), f"{name} vs fp32: in kernel {e_k:.4g}, two streams {e_s:.4g}"
        close(dq_k[rd, lo:lo + H * hs], dq_s[rd, lo:lo + H * hs], 3e-2, name + " vs the two-stream form")
    assert torch.isnan(dq_k[~rd].float()).all(), "dq / dk / dv rows at padded positions must not be written"
    # token order (o_map, g_tok): the same numbers at the mapped rows
    ntok = int(rows.sum())
    tok_of = torch.full((B * L,), -1, dtype=torch.int32)
    tok_of[rows] = torch.randperm(ntok, generator=g).to(torch.int32)
    omap = tok_of.to(dev)
    sel = tok_of[rows].long().to(dev)
    rel_t = torch.empty((B * H, L, 32), dtype=bf16, device=dev)
    o_t, lse_t = ops.flash_attn(holes.to(dev), B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel_hw=(16, ws), want_lse=True, hs_valid=hd, q_valid=qv,
                                pad_row=pad_row.to(dev), o_map=omap, o_rows=ntok, rel_table=T, rel_out=rel_t)
    assert torch.equal(o_t[sel].view(ntok, H, hd), o_k[rd].view(ntok, H, hs)[..., :hd]), "token-order o"
    do_t = torch.zeros(ntok, H * hd, dtype=bf16, device=dev)
    do_t[sel] = do_dev.to(dev)[rd].view(ntok, H, hs)[..., :hd].reshape(ntok, H * hd)
    dg = torch.full((ntok, 3 * H * hd), float("nan"), dtype=bf16, device=dev)
    ops.flash_attn_bwd(holes.to(dev), o_t, do_t, lse_t, dg, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, rel=rel_t, rel_hw=(16, ws), hs_valid=hd,
                       q_valid=qv, pad_row=pad_row.to(dev), o_map=omap, grads_tok=True, rel_table=T)
    assert not torch.isnan(dg.float()).any(), "every element of the token-order gradient must be written"
    assert torch.equal(dg[sel].view(ntok, 3, H, hd), dq_k[rd].view(ntok, 3, H, hs)[..., :hd]), "token-order dq / dk / dv"


def test_rel_bias_streams_skip_padded_positions(dev):
    """q_valid of the rel-pos streams: positions outside a window's top-left vy x vx block get no rel' row (the output starts as NaN
    and must stay NaN there), their d rel' rows are not read (NaN there) and their dq rows are not touched."""
    from grove_amd import ops
    from grove_amd.model.sam import _rcat_tables
    nb, nh, size, hd, hp = 13, 16, 14, 80, 96
    L = size * size
    g = torch.Generator().manual_seed(5)
    rel_h = (torch.randn(2 * size - 1, hd, generator=g) * 0.5).to(bf16).to(dev)
    rel_w = (torch.randn(2 * size - 1, hd, generator=g) * 0.5).to(bf16).to(dev)
    rcat, rcat_t, khp, rel_ld = _rcat_tables(size, rel_h, rel_w, hd, hp, hd ** -0.5)
    ld = 3 * nh * hp
    qkv = (torch.randn(nb * L, ld, generator=g) * 0.5).to(bf16)
    qkv.view(nb * L, 3 * nh, hp)[:, :, hd:] = 0
    shapes = [(14, 14), (4, 14), (14, 4), (4, 4)]
    valid = torch.tensor([shapes[b % 4] for b in range(nb)], dtype=torch.int32)
    mask = (torch.arange(size)[None, :, None] < valid[:, 0, None, None]) & (torch.arange(size)[None, None, :] < valid[:, 1, None, None])
    mask = mask.view(nb, L)                                        # [nb, L]
    qkv[~mask.view(-1)] = float("nan")                             # rows at padded positions are never read
    rel = torch.full((nb * nh, L, rel_ld), float("nan"), dtype=bf16, device=dev)
    ops.rel_bias_fwd(qkv.to(dev), rcat, nb, nh, L, hp, hd, out=rel, q_valid=valid.to(dev), kw=size)
    q4 = torch.nan_to_num(qkv.float()).view(nb, L, 3 * nh, hp)[:, :, :nh]
    ref = torch.einsum("bqhd,qnd->bhqn", q4, rcat.float().cpu())     # [nb, nh, L, rel_ld]
    hm = mask[:, None, :].expand(nb, nh, L)
    got = rel.float().cpu().view(nb, nh, L, rel_ld)
    close(got[hm], ref[hm], 2 ** -7, "rel' at the real positions")
    assert torch.isnan(got[~hm]).all(), "rel' rows at padded positions must not be written"
    drel = (torch.randn(nb * nh, L, rel_ld, generator=g) * 0.3).to(bf16)
    drel.view(nb, nh, L, rel_ld)[~hm] = float("nan")
    dq0 = (torch.randn(nb * L, ld, generator=g) * 0.5).to(bf16)
    dq = dq0.clone().to(dev)
    ops.rel_bias_bwd(drel.to(dev), rcat_t, dq, nb, nh, L, hp, hd, q_valid=valid.to(dev), kw=size)
    add = torch.einsum("bhqn,qdn->bqhd", torch.nan_to_num(drel.float()).view(nb, nh, L, rel_ld), rcat_t.float().cpu())
    want = dq0.float().view(nb, L, 3 * nh, hp).clone()
    want[:, :, :nh] += add * mask[:, :, None, None]
    close(dq, want.view(nb * L, ld), 2 ** -7, "dq += d rel' . Rcat at the real positions, untouched elsewhere")
    # dq_map (round 6): the same accumulation into a TOKEN-order dq with compact heads (row dq_map[(b, q)], head h at column h * hd)
    ntok = int(mask.sum())
    tok_of = torch.full((nb * L,), -1, dtype=torch.int32)
    tok_of[mask.view(-1)] = torch.randperm(ntok, generator=g).to(torch.int32)
    dt0 = (torch.randn(ntok, 3 * nh * hd, generator=g) * 0.5).to(bf16)
    dt = dt0.clone().to(dev)
    ops.rel_bias_bwd(drel.to(dev), rcat_t, dt, nb, nh, L, hp, hd, q_valid=valid.to(dev), kw=size, dq_map=tok_of.to(dev))
    want_t = dt0.float().clone().view(ntok, 3 * nh, hd)
    sel = tok_of[mask.view(-1)].long()
    want_t[sel, :nh] += add.reshape(nb * L, nh, hp)[mask.view(-1)][..., :hd]
    close(dt, want_t.view(ntok, 3 * nh * hd), 2 ** -7, "token-order dq += d rel' . Rcat")
    assert torch.equal(dt.cpu().view(ntok, 3 * nh, hd)[:, nh:], dt0.view(ntok, 3 * nh, hd)[:, nh:]), "the k / v blocks are not touched"


def test_gemm_stream_k_shape_inside_a_stream_capture(dev):
    """A shape whose eager launches take the stream-K tail, launched inside a HIP-graph capture on a stream that has no stream-K
    scratch yet. Round 3: workspaces are the caller's — ops.py hands the captured launch the shape's (already uploaded) work-list
    image and a scratch tensor from the graph's own pool, the library allocates nothing, and the replayed graph reproduces the eager
    stream-K launch bit for bit (round 2 fell back to whole tiles there)."""
    from grove_amd import _lib, ops
    L = _lib.lib()
    M, N, K = 20200, 1000, 2560
    g = torch.Generator().manual_seed(3)
    a = (torch.randn(M, K, generator=g) * 0.5).to(bf16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(bf16).to(dev)
    try:
        L.grove_gemm_set_tile_m(256)
        eager = ops.linear(a, b)                       # warm-up on the current stream: stream-K list + whole-tile list
        assert L.grove_gemm_last_stream_k() > 0
        L.grove_gemm_set_stream_k(0)
        whole = ops.linear(a, b)
        L.grove_gemm_set_stream_k(1)
        out = torch.empty_like(whole)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):                  # (captures on a fresh side stream)
            ops.linear(a, b, out=out)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), "captured stream-K launch vs eager stream-K launch"
        close(out, whole, 2 ** -7, "stream-K vs whole-tile result")
    finally:
        L.grove_gemm_set_tile_m(0)
        L.grove_gemm_set_stream_k(1)


def test_greedy_pick_is_hf_greedy_bookkeeping(dev):
    """grove_greedy_pick against the torch ops it replaces in the captured decode step: first-maximum argmax over V columns of a wider row,
    finished rows emit pad, a row finishes at eos, token / position advance, id and hidden rows filed under step = pos - pos0."""
    from grove_amd import ops
    B, V, ld, H, steps, pos0 = 3, 1000, 1024, 64, 5, 17
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(B, ld, generator=g)
    logits[:, V:] = 100.0                      # columns beyond V must be ignored
    logits[0, 123] = logits[0, 777] = 50.0     # a tie: the FIRST maximum wins
    logits[1, 2] = 60.0                        # row 1 picks eos (= 2) -> finishes
    logits = logits.to(dev)
    finished = torch.tensor([False, False, True], device=dev)   # row 2 was finished before: emits pad whatever its logits say
    tok = torch.zeros(B, dtype=torch.int32, device=dev)
    pos = torch.full((B,), pos0 + 3, dtype=torch.int32, device=dev)  # step 3
    ids_out = torch.full((B, steps), -7, dtype=torch.int64, device=dev)
    hidden = torch.randn(B, H, generator=g).to(bf16).to(dev)
    hidden32 = torch.randn(B, H, generator=g).to(dev)
    hid_out = torch.zeros(steps, B, H, dtype=bf16, device=dev)
    hid32_out = torch.zeros(steps, B, H, device=dev)
    ops.greedy_pick(logits, V, finished, tok, pos, ids_out, eos=2, pad=0, pos0=pos0, hidden=hidden, hid_out=hid_out, hidden_f32=hidden32, hid_out_f32=hid32_out)
    torch.cuda.synchronize()
    assert tok.tolist() == [123, 2, 0] and finished.tolist() == [False, True, True] and pos.tolist() == [pos0 + 4] * 3
    assert ids_out[:, 3].tolist() == [123, 2, 0] and (ids_out[:, [0, 1, 2, 4]] == -7).all()
    assert torch.equal(hid_out[3], hidden) and torch.equal(hid32_out[3], hidden32) and hid_out[[0, 1, 2, 4]].abs().sum() == 0


@pytest.mark.parametrize("Co,frames", [(640, 32), (256, 16), (1280, 16)])
def test_conv3d_temporal_tap_skipping_is_bit_identical_on_whole_tiles(dev, Co, frames):
    """Round 4: the Conv3d adapters' implicit GEMM with the temporal-padding promise (grove_gemm_params.a_frame_rows / a_frames):
    first-frame tiles skip the first tap group's K range, last-frame tiles the last one's, and the short tiles are dealt to the
    persistent blocks after the full ones. Skipped products only ever added +0.0, so every output tile that is not part of a stream-K
    split is BIT-identical to the launch without the promise; split tiles (other K cut points, fp32 sum order) agree to one bf16 ulp.
    Also against the fp32 gather reference."""
    from grove_amd import _lib, ops
    from grove_amd.model.indexing import conv3d_gather_index
    L = _lib.lib()
    T, H, W, Ci = 8, 32, 32, 128
    G = frames // T
    M = frames * H * W
    x = rnd(M, Ci, seed=21).to(dev)
    w = rnd(Co, 27 * Ci, seed=22, scale=0.03).to(dev)
    bias = rnd(Co, seed=23).to(dev)
    idx = conv3d_gather_index(G, T, H, W).to(dev)
    kw = dict(act=ops.ACT_RELU, scale_ptr=torch.tensor([0.3]).to(dev), scale_tanh=True, a_idx=idx, a_taps=27, M=M)
    try:
        L.grove_gemm_set_tile_m(256)
        L.grove_gemm_set_tap_skip(0)
        ref = ops.linear(x, w, bias, a_frames=(H * W, T), **kw)
        assert L.grove_gemm_last_variant() == 6
        split_ref = L.grove_gemm_last_stream_k()
        L.grove_gemm_set_tap_skip(1)
        out = ops.linear(x, w, bias, a_frames=(H * W, T), **kw)
        split = L.grove_gemm_last_stream_k()
        plain = ops.linear(x, w, bias, **kw)       # no promise given: the un-skipped plan
        assert torch.equal(plain, ref)
    finally:
        L.grove_gemm_set_tile_m(0)
        L.grove_gemm_set_tap_skip(1)
    same = (out == ref).all(dim=1)
    if not (split or split_ref):
        assert bool(same.all())
    else:  # rows of split tiles may differ in the last bit; every other row is identical
        frac = float(same.float().mean())
        assert frac > 0.55, frac
        assert (out.float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()
    # fp32 gather reference on a sample of rows incl. first / last frames of a group
    rows = torch.cat([torch.arange(0, 64), torch.arange(7 * H * W, 7 * H * W + 64), torch.arange(M - 64, M), torch.arange(3 * H * W + 500, 3 * H * W + 564)]).to(dev)
    src = idx[:, rows].long()                                         # [27, n]
    xg = torch.where((src >= 0)[..., None], x.float()[src.clamp_min(0)], torch.zeros(1, device=dev))   # [27, n, Ci]
    pre = torch.einsum("tnc,otc->no", xg, w.float().view(Co, 27, Ci)) + bias.float()
    want = torch.relu(pre) * math.tanh(0.3)
    close(out[rows], want, 2 ** -7, "tap-skipping conv vs fp32 gather")


@pytest.mark.parametrize("G,T,Ci,Co,split", [(2, 8, 256, 512, 0), (4, 8, 512, 1280, 0), (2, 4, 512, 1280, 1)])
def test_wgrad_conv3d_temporal_tap_skipping(dev, G, T, Ci, Co, split):
    """Round 4: the Conv3d adapters' weight gradient (one gathered TN GEMM) with the temporal-padding promise
    (grove_gemm_tn_params.b_frame_rows / b_frames): output tiles of the first tap group skip the K tiles of every group's first frame,
    tiles of the last tap group those of the last frame, and the full tiles (middle tap group) are dealt first. The skipped products
    only ever added +0.0: with whole tiles (split_tail off) the result is BIT-identical to the launch without the promise; with a cut
    tail (fp32 atomics, other cut points) it agrees to fp32 sum-order noise. Also against the 128 x 128 kernel."""
    from grove_amd import _lib, ops
    from grove_amd.model.indexing import conv3d_gather_index
    L = _lib.lib()
    H = W = 32
    Mtok = G * T * H * W
    xx, dz = rnd(Mtok, Ci, seed=31).to(dev), rnd(Mtok, Co, seed=32).to(dev)
    idx = conv3d_gather_index(G, T, H, W).to(dev)
    z = lambda: torch.zeros(Co, 27 * Ci, dtype=torch.float32, device=dev)
    try:
        L.grove_gemm_tn_set_pipelined(1)
        L.grove_gemm_tn_set_split_tail(split)
        L.grove_gemm_tn_set_tap_skip(0)
        ref = ops.wgrad(dz, xx, z(), b_idx=idx, b_taps=27, b_frames=(H * W, T))
        assert L.grove_gemm_tn_last_skip() == 0
        L.grove_gemm_tn_set_tap_skip(1)
        out = ops.wgrad(dz, xx, z(), b_idx=idx, b_taps=27, b_frames=(H * W, T))
        assert L.grove_gemm_tn_last_skip() == 1
        plain = ops.wgrad(dz, xx, z(), b_idx=idx, b_taps=27)      # no promise: no skipping
        assert L.grove_gemm_tn_last_skip() == 0
        if not split:
            assert torch.equal(plain, ref)
            assert torch.equal(out, ref), (out - ref).abs().max().item()
        else:
            close(out, ref, 2e-6, "tap-skipping wgrad with a cut tail")
        # accumulation onto existing values, twice (the optimizer's gradient buffer is accumulated into)
        acc = ops.wgrad(dz, xx, out.clone(), b_idx=idx, b_taps=27, b_frames=(H * W, T))
        close(acc, 2 * ref, 2e-6, "accumulating launch")
        L.grove_gemm_tn_set_pipelined(0)
        small = ops.wgrad(dz, xx, z(), b_idx=idx, b_taps=27)
        close(out, small, 3e-6, "tap-skipping wgrad vs the 128 x 128 kernel")
    finally:
        L.grove_gemm_tn_set_pipelined(-1)
        L.grove_gemm_tn_set_split_tail(1)
        L.grove_gemm_tn_set_tap_skip(1)


@pytest.mark.parametrize("B,H,hs,Lq,Lk,ragged", [(2, 4, 128, 54, 703, False), (3, 2, 64, 130, 300, True), (1, 4, 128, 1, 200, False), (2, 4, 32, 200, 333, True)])
def test_flash_attention_tail_queries(dev, B, H, hs, Lq, Lk, ragged):
    """The general fused kernels with Lq != Lk: causal attention of the LAST Lq positions against all Lk keys (bottom-right aligned
    mask: query i is position Lk - Lq + i; per-sequence kv_len for right padding) — what LlamaStack's last layer runs when only the
    answer's rows are consumed (round 4). Forward and backward against torch autograd in fp32."""
    from grove_amd import ops
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(B * Lq, H * hs, generator=g) * 0.7).to(bf16)
    kv = (torch.randn(B * Lk, 2 * H * hs, generator=g) * 0.7).to(bf16)
    do = torch.randn(B * Lq, H * hs, generator=g).to(bf16)
    kv_len = torch.tensor([Lk - 7 * b for b in range(B)], dtype=torch.int32) if ragged else None
    alpha = hs ** -0.5
    out, lse = ops.flash_attn_tail(q.to(dev), kv.to(dev), B, Lq, Lk, H, hs, alpha, kv_len=kv_len.to(dev) if ragged else None, want_lse=True)
    qr = q.float().view(B, Lq, H, hs).permute(0, 2, 1, 3).requires_grad_(True)
    kr = kv.float()[:, :H * hs].reshape(B, Lk, H, hs).permute(0, 2, 1, 3).requires_grad_(True)
    vr = kv.float()[:, H * hs:].reshape(B, Lk, H, hs).permute(0, 2, 1, 3).requires_grad_(True)
    s = (qr @ kr.transpose(-1, -2)) * alpha
    i = torch.arange(Lq)[:, None] + (Lk - Lq)
    j = torch.arange(Lk)[None, :]
    mask = (j > i)[None, None].expand(B, H, Lq, Lk).clone()
    if ragged:
        for b in range(B):
            mask[b, :, :, int(kv_len[b]):] = True
    valid_q = ~mask.all(-1)                                     # (a padded query past kv_len sees nothing: its row is unused)
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    o_ref = p @ vr
    got = out.float().cpu().view(B, Lq, H, hs).permute(0, 2, 1, 3)
    sel = valid_q[..., None].expand_as(o_ref)
    close(got[sel], o_ref.detach()[sel], 1.5e-2, "tail attention forward")
    lse_ref = torch.logsumexp(s, -1)
    assert (lse.float().cpu().view(B, H, Lq)[valid_q] - lse_ref.detach()[valid_q]).abs().max().item() < 2e-2
    dor = do.float().view(B, Lq, H, hs).permute(0, 2, 1, 3) * valid_q[..., None]
    (o_ref * dor).sum().backward()
    dq = torch.empty_like(q).to(dev)
    dkv = torch.empty_like(kv).to(dev)
    do_dev = (dor.permute(0, 2, 1, 3).reshape(B * Lq, H * hs)).to(bf16).to(dev)
    ops.flash_attn_tail_bwd(q.to(dev), kv.to(dev), out, do_dev, lse, dq, dkv, B, Lq, Lk, H, hs, alpha, kv_len=kv_len.to(dev) if ragged else None)
    dq_ref = qr.grad.permute(0, 2, 1, 3).reshape(B * Lq, H * hs)
    dk_ref = kr.grad.permute(0, 2, 1, 3).reshape(B * Lk, H * hs)
    dv_ref = vr.grad.permute(0, 2, 1, 3).reshape(B * Lk, H * hs)
    vq = valid_q.permute(0, 2, 1).reshape(B * Lq, H)[..., None].expand(-1, -1, hs).reshape(B * Lq, H * hs)
    close(dq.float().cpu()[vq], dq_ref[vq], 2.5e-2, "tail attention dq")
    close(dkv.float().cpu()[:, :H * hs], dk_ref, 2.5e-2, "tail attention dk")
    close(dkv.float().cpu()[:, H * hs:], dv_ref, 2.5e-2, "tail attention dv")


@pytest.mark.parametrize("B,H,hs,L", [(2, 4, 128, 703), (2, 2, 64, 200), (1, 4, 32, 130)])
def test_flash_attention_bwd_fused_inverse_rope(dev, B, H, hs, L):
    """Round 4: the inverse RoPE of dq | dk inside the epilogues of the two backward kernels (grove_flash_attn_params.rope) against the
    unfused sequence (backward, then grove_rope_inplace(inverse)): the fused form rotates the fp32 accumulators before the one bf16
    rounding, the unfused one rounds, rotates and rounds again — equal to bf16 rounding; dv is untouched (bit-identical)."""
    from grove_amd import ops
    theta = 10000.0
    g = torch.Generator().manual_seed(9)
    qkv = (torch.randn(B * L, 3 * H * hs, generator=g) * 0.6).to(bf16).to(dev)
    pos = torch.arange(L, dtype=torch.int32).repeat(B).to(dev)
    ops.rope_(qkv, pos, 0, 2 * H, hs, theta)                      # q | k rotated, as the forward leaves them
    do = torch.randn(B * L, H * hs, generator=g).to(bf16).to(dev)
    alpha = hs ** -0.5
    out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=True, want_lse=True)
    ref = torch.empty_like(qkv)
    ops.flash_attn_bwd(qkv, out, do, lse, ref, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=True)
    ops.rope_(ref, pos, 0, 2 * H, hs, theta, inverse=True)
    got = torch.empty_like(qkv)
    table = ops.rope_table(hs, theta, L + 5, dev)
    ops.flash_attn_bwd(qkv, out, do, lse, got, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=True, rope=table)
    assert torch.equal(got[:, 2 * H * hs:], ref[:, 2 * H * hs:])   # dv
    close(got[:, :2 * H * hs], ref[:, :2 * H * hs], 2 ** -6, "fused inverse RoPE of dq | dk")
    # the table is HF's: cos / sin of pos * theta^(-2 i / hd) in fp32
    inv = 1.0 / (theta ** (torch.arange(0, hs, 2, dtype=torch.float32) / hs))
    ang = torch.arange(L + 5, dtype=torch.float32)[:, None] * inv[None]
    # (fp32 angles up to ~700 rad: the device's argument reduction and the host's differ by ~1e-4 there; bf16 rounds at 4e-3)
    assert (table.cpu() - torch.cat([ang.cos(), ang.sin()], 1)).abs().max().item() < 1e-3


def test_rope_from_table_matches_evaluated_rope(dev):
    """grove_rope_inplace with the cos | sin table (round 4: two 32-byte reads per thread instead of 8 x (powf + sincosf)) against the
    evaluated form, forward and inverse: equal to bf16 rounding (the angles differ by fp32 argument-reduction noise only)."""
    from grove_amd import ops
    rows, H, hd, theta = 1406, 8, 128, 10000.0
    x = rnd(rows, 3 * H * hd, seed=41).to(dev)
    pos = (torch.arange(rows, dtype=torch.int32) % 703).to(dev)
    table = ops.rope_table(hd, theta, 703, dev)
    for inv in (False, True):
        a, b = x.clone(), x.clone()
        ops.rope_(a, pos, 0, 2 * H, hd, theta, inverse=inv)
        ops.rope_(b, pos, 0, 2 * H, hd, theta, inverse=inv, table=table)
        assert torch.equal(a[:, 2 * H * hd:], b[:, 2 * H * hd:])          # v untouched
        assert (a.float() - b.float()).abs().max().item() <= 2 ** -7 * a.float().abs().max().item()
        assert float((a != b).float().mean()) < 0.02                        # all but a few elements round identically
