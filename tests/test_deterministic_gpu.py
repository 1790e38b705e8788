"""Deterministic mode (include/grove_hip.h: grove_set_deterministic): the sums that meet in global memory through fp32 atomics get a
fixed order, so a program gives the same BITS run after run. Each ticketed kernel is run several times on inputs whose many blocks
collide on a few addresses (the worst case for arrival order), the whole-step form is tests/test_train_gpu.py::test_training_steps_repeat_bit_for_bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from grove_amd import ops  # noqa: E402


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture
def det(dev):
    prev = ops.set_deterministic(True)
    yield
    ops.set_deterministic(prev)


def _repeat(fn, n=4):
    outs = []
    for _ in range(n):
        r = fn()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (r if isinstance(r, (tuple, list)) else (r,))])
    return outs


def _all_equal(outs):
    return all(torch.equal(a, b) for o in outs[1:] for a, b in zip(outs[0], o))


def _gen(dev, seed=0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return g


def test_mode_switch(dev):
    prev = ops.set_deterministic(True)
    assert ops.set_deterministic(False) is True
    assert ops.set_deterministic(prev) is False


@pytest.mark.parametrize("rows,C,n_dst", [(6000, 64, 7), (2812, 4096, 40)])
def test_scatter_add_is_ordered(dev, det, rows, C, n_dst):
    g = _gen(dev)
    src = torch.randn(rows, C, device=dev, generator=g).bfloat16()
    idx = torch.randint(-1, n_dst, (rows,), device=dev, generator=g, dtype=torch.int32)
    outs = _repeat(lambda: ops.scatter_add_f32(src, torch.zeros(n_dst, C, device=dev), idx, rows, C))
    assert _all_equal(outs)
    # the serial definition: rows in order (fp32 adds)
    ref = torch.zeros(n_dst, C)
    s, ix = src.float().cpu(), idx.cpu()
    for r in range(rows if C <= 64 else 300):
        if ix[r] >= 0:
            ref[ix[r]] += s[r]
    if C <= 64:
        assert torch.equal(outs[0][0].cpu(), ref)
    src32 = src.float()
    outs = _repeat(lambda: ops.scatter_add_rows_f32(src32, torch.zeros(n_dst, C, device=dev), idx))
    assert _all_equal(outs)


def test_block_sums_are_ordered(dev, det):
    g = _gen(dev, 1)
    # cross entropy: one add per row
    R, V = 3000, 1000
    logits = (torch.randn(R, V, device=dev, generator=g) * 3).bfloat16()
    labels = torch.randint(0, V, (R,), device=dev, generator=g, dtype=torch.int32)
    outs = _repeat(lambda: ops.cross_entropy(logits, labels, V))
    assert _all_equal(outs)
    ref = torch.nn.functional.cross_entropy(logits.float(), labels.long(), reduction="sum")
    assert abs(outs[0][0].item() - ref.item()) <= 1e-4 * abs(ref.item())
    # sum of squares over 2048 blocks, dot over 2048 blocks
    x = torch.randn(5_000_000, device=dev, generator=g)
    outs = _repeat(lambda: ops.sumsq(x))
    assert _all_equal(outs) and abs(outs[0][0].item() - (x.double() ** 2).sum().item()) <= 1e-4 * x.numel()
    a, b = x[:4_000_000].bfloat16(), x[1_000_000:].bfloat16()
    outs = _repeat(lambda: ops.dot(a, b, torch.zeros(1, device=dev)))
    assert _all_equal(outs)
    # column sums
    y = torch.randn(5000, 640, device=dev, generator=g).bfloat16()
    outs = _repeat(lambda: ops.colsum(y))
    assert _all_equal(outs) and torch.allclose(outs[0][0], y.float().sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("rms", [False, True])
def test_norm_dweight_is_ordered(dev, det, rms):
    g = _gen(dev, 2)
    rows, C = 4099, 1280
    x = torch.randn(rows, C, device=dev, generator=g).bfloat16()
    dy = torch.randn(rows, C, device=dev, generator=g).bfloat16()
    w = (1 + 0.1 * torch.randn(C, device=dev, generator=g)).bfloat16()
    xf = x.float()
    mean, rstd = xf.mean(1), (xf.var(1, unbiased=False) + 1e-6).rsqrt()

    def run():
        dw = torch.zeros(C, device=dev)
        db = torch.zeros(C, device=dev)
        if rms:
            ops.rmsnorm_bwd(x, w, dy, 1e-6, dweight=dw)
        else:
            ops.layernorm_bwd(x, w, dy, mean, rstd, dweight=dw, dbias=db)
        return dw, db

    outs = _repeat(run)
    assert _all_equal(outs)
    if not rms:
        assert torch.allclose(outs[0][1], dy.float().sum(0), rtol=1e-4, atol=1e-2)


def test_weight_gradient_gemms_run_whole_k(dev, det):
    """The shapes the default mode splits over K (few tiles, long K) and the cut tail of the pipelined TN kernel."""
    g = _gen(dev, 3)
    for (K, M, N) in [(8192, 256, 256), (4096, 64, 1280), (2048, 1280 * 3, 1280 * 7 + 256)]:
        dy = torch.randn(K, M, device=dev, generator=g).bfloat16()
        x = torch.randn(K, N, device=dev, generator=g).bfloat16()
        outs = _repeat(lambda: ops.wgrad(dy, x, torch.zeros(M, N, device=dev)), n=3)
        assert _all_equal(outs), (K, M, N)
        ref = dy.float().t() @ x.float()
        assert (outs[0][0] - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    # NT form with an accumulating fp32 C (split-K in the default mode)
    A = torch.randn(64, 16384, device=dev, generator=g).bfloat16()
    B = torch.randn(128, 16384, device=dev, generator=g).bfloat16()

    def nt():
        Cf = torch.zeros(64, 128, device=dev)
        ops.gemm_raw(A, B, Cf, 64, 128, 16384, 16384, 16384, 128, accumulate=True)
        return Cf

    assert _all_equal(_repeat(nt, n=3))


def test_decoder_small_kernels_are_ordered(dev, det):
    g = _gen(dev, 4)
    # few-key attention backward: dK / dV summed over the query blocks (both kernels: d = 16 vectorised and the generic one)
    for (inst, heads, d, Lq, Lk) in [(3, 8, 16, 1024, 7), (2, 4, 32, 200, 6)]:
        HD = heads * d
        q = torch.randn(inst * Lq, HD, device=dev, generator=g).bfloat16()
        k = torch.randn(inst * Lk, HD, device=dev, generator=g).bfloat16()
        v = torch.randn(inst * Lk, HD, device=dev, generator=g).bfloat16()
        o = ops.small_attn(q, k, v, inst, heads, d, Lq, Lk)
        do = torch.randn(inst * Lq, HD, device=dev, generator=g).bfloat16()
        outs = _repeat(lambda: ops.small_attn_bwd(q, k, v, o, do, inst, heads, d, Lq, Lk))
        assert _all_equal(outs), (inst, heads, d, Lq, Lk)
    # box losses: one add per block of 256 instances
    N = 5000
    pb = torch.rand(N, 4, device=dev, generator=g) * 0.5 + 0.2
    gb = torch.rand(N, 4, device=dev, generator=g) * 0.5 + 0.2
    ol = torch.randn(N, device=dev, generator=g)
    vis = (torch.rand(N, device=dev, generator=g) > 0.3).float()
    outs = _repeat(lambda: ops.box_losses(pb, ol, gb, vis, 1.0 / N, 1.0 / N)[0])
    assert _all_equal(outs)
    # box / objectness heads: every instance (block) adds to the same weight gradients
    N, D = 300, 256
    x = torch.randn(N, D, device=dev, generator=g)
    W1 = (torch.randn(D, D, device=dev, generator=g) * 0.05).bfloat16()
    b1 = torch.zeros(D, device=dev).bfloat16()
    W2 = (torch.randn(4, D, device=dev, generator=g) * 0.05).bfloat16()
    b2 = torch.zeros(4, device=dev).bfloat16()
    Wo = (torch.randn(1, D, device=dev, generator=g) * 0.05).bfloat16()
    bo = torch.zeros(1, device=dev).bfloat16()
    box, obj, hidden = ops.box_head(x, W1, b1, W2, b2, Wo, bo)
    dbox = torch.randn(N, 4, device=dev, generator=g)
    dobj = torch.randn(N, device=dev, generator=g)

    def heads():
        gr = {"dW1": torch.zeros(D, D, device=dev), "db1": torch.zeros(D, device=dev), "dW2": torch.zeros(4, D, device=dev),
              "db2": torch.zeros(4, device=dev), "dWo": torch.zeros(D, device=dev), "dbo": torch.zeros(1, device=dev)}
        ops.box_head_bwd(x, W1, W2, Wo, hidden, box, dbox, dobj, gr)
        return [gr[k] for k in sorted(gr)]

    assert _all_equal(_repeat(heads))
