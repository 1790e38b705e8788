"""The Winograd F(2x2x2, 3x3x3) form of the Conv3d adapters (csrc/winograd.hip + the grouped NT / K-batched TN GEMMs) through the C-ABI:
every transform against the fp32 restatement in oracle/winograd_ref.py, the assembled forward / dgrad / wgrad against F.conv3d and its
autograd (the reference's arithmetic: image_encoder.py:43-59) and against the 27-tap implicit-GEMM kernels they replace, and the SAM
tower with the switch on against the switch off."""
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
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    err = (out - ref).abs().max().item()
    lim = rtol * max(ref.abs().max().item(), 1e-6)
    assert err <= lim, f"{what}: max abs err {err:.4g} > {lim:.4g}"


def rel_rms(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("geom,C", [((1, 2, 4, 6), 66), ((2, 4, 6, 10), 128), ((1, 8, 16, 16), 320)])
def test_token_transforms_match_the_restatement(dev, geom, C):
    """B^T-transform of the zero-padded overlapping input tiles and A-transform of the output-gradient tiles: every tile, every point,
    ragged channel counts (the last block of channel pairs is partly idle), a volume whose every tile touches a face."""
    from grove_amd import ops
    from oracle import winograd_ref as R
    g, T, H, W = geom
    x = rnd(g * T * H * W, C, seed=3)
    x5 = R.tokens_to_5d(x.float(), geom)
    for mode, ref in ((0, R.input_transform(x5)), (1, R.grad_transform(x5))):
        out = ops.wino3d_transform_tokens(x.to(dev), geom, mode)
        want = R.to_point_major(ref)
        # the kernel's fp32 sums associate differently from einsum's: equal up to one bf16 rounding of the result
        close(out, want, 2.0 ** -8, f"token transform mode {mode}")
        assert rel_rms(out.float(), R.r16(want)) < 2e-3


def test_framed_tokens_and_padded_points(dev):
    """CLIP's layout: frames of 1 CLS row + H W patch rows (frame_rows, row_offset) and a tile count that is not a multiple of the grouped
    GEMM's 256-row tiles (tiles_ld): the whole convolution against F.conv3d, the CLS rows of the output untouched."""
    from grove_amd import ops
    from oracle import winograd_ref as R
    geom, C = (1, 8, 6, 12), 128          # 4 * 3 * 6 = 72 tiles -> padded to 256 per point
    g, T, H, W = geom
    fr = H * W + 1
    x = rnd(g * T * fr, C, seed=41)
    w = rnd(C, C, 3, 3, 3, seed=42, scale=0.05)
    bias = rnd(C, seed=43)
    wp = w.permute(0, 2, 3, 4, 1).reshape(C, 27 * C).contiguous()
    patches = x.float().reshape(g * T, fr, C)[:, 1:].reshape(g * T * H * W, C)
    ref = F.conv3d(R.tokens_to_5d(patches, geom), w.float(), bias.float(), padding=1).permute(0, 2, 3, 4, 1).reshape(g * T, H * W, C)
    want = x.float().reshape(g * T, fr, C).clone()
    want[:, 1:] = torch.relu(ref) * math.tanh(0.3) + want[:, 1:]
    V = ops.wino3d_transform_tokens(x.to(dev), geom, 0, frames=(fr, 1), tiles_ld=256)
    close(V[:, :72], R.to_point_major(R.input_transform(R.tokens_to_5d(patches, geom))), 2.0 ** -8, "framed input transform")
    out = x.to(dev).clone()
    ops.wino3d_conv(x.to(dev), ops.wino3d_transform_weight(wp.to(dev)), geom, out, bias=bias.to(dev), act=ops.ACT_RELU,
                    scale_ptr=torch.tensor([0.3]).to(dev), scale_tanh=True, residual=x.to(dev), frames=(fr, 1))
    close(out, want.reshape(-1, C), 1.2e-2, "framed winograd adapter vs conv3d")
    assert torch.equal(out.cpu().reshape(g * T, fr, C)[:, 0], x.reshape(g * T, fr, C)[:, 0])


def test_weight_transform_and_its_adjoint(dev):
    from grove_amd import ops
    from oracle import winograd_ref as R
    Co, Ci = 24, 66
    w = rnd(Co, Ci, 3, 3, 3, seed=5, scale=0.05)
    wp = w.permute(0, 2, 3, 4, 1).reshape(Co, 27 * Ci).contiguous()
    U = ops.wino3d_transform_weight(wp.to(dev))
    close(U, R.weight_to_point_major(R.weight_transform(w.float())), 2.0 ** -8, "weight transform")
    # G^T-transform of a point-major fp32 tensor, accumulated with the tanh(alpha) scale into the tap-major gradient
    dU = torch.randn(64, Co, Ci, generator=torch.Generator().manual_seed(6))
    g0 = torch.randn(Co, 27 * Ci, generator=torch.Generator().manual_seed(7))
    gw = g0.clone().to(dev)
    ops.wino3d_wgrad_output(dU.to(dev), gw, scale_ptr=torch.tensor([0.3]).to(dev), scale_tanh=True)
    dU5 = dU.reshape(4, 4, 4, Co, Ci).permute(3, 4, 0, 1, 2)
    want = g0 + math.tanh(0.3) * R.t3(R.G.t().contiguous(), dU5).permute(0, 2, 3, 4, 1).reshape(Co, 27 * Ci)
    close(gw, want, 2e-6, "wgrad output transform")


def test_output_transform_epilogue(dev):
    from grove_amd import ops
    from oracle import winograd_ref as R
    geom, C = (1, 4, 6, 8), 66
    g, T, H, W = geom
    tiles = ops.wino3d_tiles(geom)
    Mh = rnd(64, tiles, C, seed=8)
    bias, res = rnd(C, seed=9), rnd(g * T * H * W, C, seed=10)
    M5 = Mh.float().reshape(4, 4, 4, g, T // 2, H // 2, W // 2, C).permute(3, 7, 4, 5, 6, 0, 1, 2)
    y = R.untile_out(R.t3(R.AT, M5)).permute(0, 2, 3, 4, 1).reshape(-1, C) + bias.float()
    out = torch.empty(g * T * H * W, C, dtype=bf16, device=dev)
    aux = torch.empty_like(out)
    ops.wino3d_output(Mh.to(dev), geom, out, bias=bias.to(dev), act=ops.ACT_RELU, scale_ptr=torch.tensor([0.2]).to(dev), scale_tanh=True,
                      residual=res.to(dev), aux=aux)
    close(aux, y, 2.0 ** -8, "pre-activation")
    close(out, torch.relu(y) * math.tanh(0.2) + res.float(), 2.0 ** -8, "output transform + epilogue")
    plain = torch.empty_like(out)
    ops.wino3d_output(Mh.to(dev), geom, plain)
    close(plain, y - bias.float(), 2.0 ** -8, "output transform, no epilogue")


@pytest.mark.parametrize("groups,rows,N,K", [(4, 256, 136, 128), (64, 512, 320, 320)])
def test_grouped_b_gemm(dev, groups, rows, N, K):
    """grove_gemm_params.b_group_rows: group g of the A rows times ITS weight matrix, one launch (the 64 transform points)."""
    from grove_amd import ops
    a = rnd(groups * rows, K, seed=11)
    b = rnd(groups, N, K, seed=12, scale=0.1)
    out = torch.empty(groups * rows, N, dtype=bf16, device=dev)
    ops.gemm_raw(a.to(dev), b.to(dev), out, groups * rows, N, K, K, K, N, b_group=rows)
    ref = torch.einsum("grk,gnk->grn", a.float().reshape(groups, rows, K), b.float()).reshape(groups * rows, N)
    close(out, ref, 6e-3, "grouped B")
    with pytest.raises(RuntimeError):  # groups must be whole 256-row tiles
        ops.gemm_raw(a.to(dev), b.to(dev), out, groups * rows, N, K, K, K, N, b_group=rows - 64)


@pytest.mark.parametrize("batches,K,M,N,overwrite", [(3, 128, 64, 72, True), (64, 256, 320, 320, True), (5, 192, 264, 136, False)])
def test_k_batched_tn_gemm(dev, batches, K, M, N, overwrite):
    """grove_gemm_tn_params.k_batches: batch b = rows [b K, (b + 1) K) of both stacked operands -> its own [M, N] product."""
    from grove_amd import ops
    dy, x = rnd(batches * K, M, seed=13), rnd(batches * K, N, seed=14)
    g0 = torch.randn(batches, M, N, generator=torch.Generator().manual_seed(15))
    out = g0.clone().to(dev)
    ops.wgrad(dy.to(dev), x.to(dev), out, K=K, k_batches=batches, sC_batch=M * N, overwrite=overwrite, M=M, N=N, alpha=0.5)
    ref = 0.5 * torch.einsum("bkm,bkn->bmn", dy.float().reshape(batches, K, M), x.float().reshape(batches, K, N))
    close(out, ref if overwrite else g0 + ref, 3e-5, "k-batched TN")


def _conv_case(geom, Ci, Co, seed):
    g, T, H, W = geom
    rows = g * T * H * W
    x = rnd(rows, Ci, seed=seed)
    w = rnd(Co, Ci, 3, 3, 3, seed=seed + 1, scale=0.05)
    bias = rnd(Co, seed=seed + 2)
    wp = w.permute(0, 2, 3, 4, 1).reshape(Co, 27 * Ci).contiguous()
    return rows, x, w, bias, wp


@pytest.mark.parametrize("geom,C", [((1, 8, 16, 16), 128), ((2, 8, 16, 16), 320)])
def test_winograd_conv_forward_vs_conv3d_and_direct_kernel(dev, geom, C):
    """tanh(alpha) relu(Conv3d(x) + b) + x: the Winograd pipeline against F.conv3d in fp32 (the reference's arithmetic) and against the
    27-tap implicit GEMM it replaces. Tolerance = the direct kernel's (bf16 output rounding) + the two extra operand roundings, which the
    fp32 accumulation over C averages down (CPU study: profiles/r06_winograd_study_deep_narrow.json)."""
    from grove_amd import ops
    from grove_amd.model.indexing import conv3d_gather_index
    from oracle import winograd_ref as R
    rows, x, w, bias, wp = _conv_case(geom, C, C, 21)
    a = torch.tensor([0.3])
    ref = F.conv3d(R.tokens_to_5d(x.float(), geom), w.float(), bias.float(), padding=1).permute(0, 2, 3, 4, 1).reshape(rows, C)
    want = torch.relu(ref) * math.tanh(0.3) + x.float()
    xd = x.to(dev)
    out, pre = torch.empty_like(xd), torch.empty_like(xd)
    U = ops.wino3d_transform_weight(wp.to(dev))
    _, V = ops.wino3d_conv(xd, U, geom, out, bias=bias.to(dev), act=ops.ACT_RELU, scale_ptr=a.to(dev), scale_tanh=True, residual=xd, aux=pre, keep_V=True)
    close(pre, ref, 1.2e-2, "winograd pre-activation vs conv3d")
    close(out, want, 1.2e-2, "winograd adapter vs conv3d")
    idx = conv3d_gather_index(*geom).to(dev)
    direct = ops.linear(xd, wp.to(dev), bias.to(dev), act=ops.ACT_RELU, residual=xd, scale_ptr=a.to(dev), scale_tanh=True, a_idx=idx, a_taps=27, M=rows)
    # on N(0, 1) inputs the direct kernel's only error is the bf16 rounding of its output (1.3e-3 rms of the adapter output here); the
    # Winograd form adds the rounding of V, U and of the 64 products: 3x that on the conv term alone (which the model scales by
    # tanh(alpha) = 0.1 before it meets a stream that is itself rounded to bf16 every block)
    assert rel_rms(out.float(), want) < 4 * max(rel_rms(direct.float(), want), 1e-3)
    # the restatement with the same rounding points (operands and products in bf16) agrees to a product-rounding's worth
    y16, _ = R.wino_conv(R.tokens_to_5d(x.float(), geom), w.float(), m16=True)
    close(pre, y16.permute(0, 2, 3, 4, 1).reshape(rows, C) + bias.float(), 6e-3, "winograd pre-activation vs the fake-quantised restatement")
    assert V.shape == (64, ops.wino3d_tiles(geom), C)


@pytest.mark.parametrize("geom,C", [((1, 8, 16, 16), 128), ((2, 8, 16, 16), 384)])
def test_winograd_wgrad_and_dgrad_vs_autograd(dev, geom, C):
    from grove_amd import ops
    from grove_amd.model.indexing import conv3d_gather_index
    from oracle import winograd_ref as R
    rows, x, w, bias, wp = _conv_case(geom, C, C, 31)
    dz = rnd(rows, C, seed=35)
    wf = w.float().requires_grad_(True)
    x5 = R.tokens_to_5d(x.float(), geom).requires_grad_(True)
    y = F.conv3d(x5, wf, padding=1).permute(0, 2, 3, 4, 1).reshape(rows, C)
    y.backward(dz.float())
    ref_w = wf.grad.permute(0, 2, 3, 4, 1).reshape(C, 27 * C)
    ref_x = x5.grad.permute(0, 2, 3, 4, 1).reshape(rows, C)
    a = torch.tensor([0.3]).to(dev)
    s = math.tanh(0.3)
    xd, dzd = x.to(dev), dz.to(dev)
    V = ops.wino3d_transform_tokens(xd, geom, 0)
    g0 = torch.randn(C, 27 * C, generator=torch.Generator().manual_seed(36)) * ref_w.abs().max() * 0.1
    gw = g0.clone().to(dev)
    ops.wino3d_wgrad(dzd, V, geom, gw, scale_ptr=a, scale_tanh=True)
    close(gw, g0 + s * ref_w, 6e-3, "winograd wgrad vs autograd")
    idx = conv3d_gather_index(*geom).to(dev)
    direct = torch.zeros(C, 27 * C, dtype=torch.float32, device=dev)
    ops.wgrad(dzd, xd, direct, b_idx=idx, b_taps=27, scale_ptr=a, scale_tanh=True)
    cos = F.cosine_similarity((gw.cpu() - g0).flatten().double(), direct.cpu().flatten().double(), dim=0).item()
    assert cos > 0.9999, cos  # (the direct form multiplies the bf16 operands exactly; here every product carries two fresh roundings)
    # dgrad = the same pipeline on the output gradient with the flipped, channel-swapped taps; + dy rides in the output transform
    wd = torch.empty(C, 27 * C, dtype=bf16, device=dev)
    ops.transpose(wp.to(dev), C, C, 27 * C, wd[:, 26 * C:], 27 * C, batch=(27, 1), s_in=(C, 0), s_out=(-C, 0))
    dy = rnd(rows, C, seed=37).to(dev)
    dx = torch.empty(rows, C, dtype=bf16, device=dev)
    ops.wino3d_conv(dzd, ops.wino3d_transform_weight(wd), geom, dx, scale_ptr=a, scale_tanh=True, residual=dy)
    close(dx, dy.float().cpu() + s * ref_x, 1.2e-2, "winograd dgrad vs autograd")


def test_sam_tower_winograd_vs_direct_adapters(dev):
    """The SAM tower (tiny dims, real 512-pixel frames, alpha = 0.1) forward + backward with the adapters in Winograd form against the
    same tower on the 27-tap kernels: embeddings and every adapter gradient. With only the WEIGHT GRADIENT in Winograd form the forward
    and the gradient flow are bit-identical, so its gradients must agree to the operand rounding; with forward and dgrad switched too
    the ReLU masks and everything downstream move by a bf16 rounding's worth, as between any two bf16 implementations."""
    from grove_amd.model.sam import SamEncoder
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    d = TINY
    sd = synthetic_state_dict(d, device=dev, dtype=bf16)
    batch = synthetic_batch(d, B=1, T=8, L=40, n_det=2, seed=3)
    img = batch.as_kwargs()["grounding_enc_images"].to(dev).to(bf16)
    names = [k for k in sd if ".image_encoder.adapters." in k]
    sdp = dict(sd)
    for j in range(len(d.sam_global)):  # the tower wants the adapter weights stored tap-major (GROVEForCausalLM packs them)
        k = f"model.grounding_encoder.image_encoder.adapters.{j}.conv3d.weight"
        sdp[k] = sd[k].permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    res = {}
    for arm, wino in (("direct", set()), ("wgrad", {"wgrad"}), ("all", {"fwd", "dgrad", "wgrad"})):
        grads = {k: torch.zeros(sd[k].numel(), dtype=torch.float32, device=dev) for k in names}
        sam = SamEncoder(sdp, d, dev, train=True, grads=grads)
        sam.wino = wino
        assert (sam._wino_geom(8 * d.sam_grid ** 2, d.sam_dim) is not None) == bool(wino)
        out, saved = sam.forward(img, save=True)
        d_out = rnd(out.shape[0] * out.shape[1], out.shape[2], seed=5).to(dev)
        sam.backward(saved, d_out)
        torch.cuda.synchronize()
        res[arm] = (out.float().cpu(), {k: v.cpu().clone() for k, v in grads.items()})
    a = res["direct"]
    assert torch.equal(res["wgrad"][0], a[0])
    assert rel_rms(res["all"][0], a[0]) < 6e-3, rel_rms(res["all"][0], a[0])
    for arm, lim in (("wgrad", 0.9999), ("all", 0.995)):
        b = res[arm]
        for k in a[1]:
            if float(a[1][k].norm()) == 0:
                continue
            if k.endswith("alpha"):  # one number = a sum of 5e5 signed terms that cancel under a random d_out (and meet through fp32 atomics): no ruler
                continue
            if arm == "wgrad" and not k.endswith("conv3d.weight"):
                assert torch.allclose(a[1][k], b[1][k], rtol=1e-4, atol=1e-5 * float(a[1][k].abs().max())), k  # (fp32 atomics: same terms, any order)
                continue
            cos = F.cosine_similarity(a[1][k].double().flatten(), b[1][k].double().flatten(), dim=0).item()
            assert cos > lim, (arm, k, cos)
            assert 0.98 < float(b[1][k].norm() / a[1][k].norm()) < 1.02, (arm, k)
