"""Out-of-bounds WRITE screen for the kernels whose tiles overhang their problem (round 5: the GEMV's padded-row stores were found by a
fault that only happened when `y` ended a mapped segment — no GPU sanitizer exists on this pool). Every tensor a wrapper allocates during
the call is carved out of a larger buffer whose margins hold a byte pattern; after the call the margins must be intact; the inside of
every `torch.empty` tensor starts as NaN (-1 for integers), so an element that is read without ever having been written shows up as a
non-finite result. Values are checked elsewhere (tests/test_kernels_gpu.py); here only that nothing outside the tensors was touched
and nothing inside them was consumed unwritten."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16
PAT = 0xA5
MARGIN = 8192  # bytes on each side (a 64-row tile of 128 bf16 columns overhanging by a few rows lands inside it)


class GuardedAllocs:
    """Replaces torch.empty / torch.zeros for CUDA tensors while active."""

    def __init__(self, poison=False):
        # poison: the INSIDE of every torch.empty tensor starts as all-ones bytes (NaN as bf16 / fp16 / fp32, -1 as an integer), so an
        # element a kernel never writes but a later one reads turns the results non-finite instead of passing on whatever was there
        self.bufs = []
        self.poison = poison
        self._empty, self._zeros = torch.empty, torch.zeros

    def carve(self, shape, dtype, device, zero=False):
        n = int(math.prod(shape))
        es = torch.empty((), dtype=dtype).element_size()
        m = MARGIN // es
        raw = self._empty(n + 2 * m, dtype=dtype, device=device)
        raw.view(torch.uint8).fill_(PAT)
        t = raw[m:m + n]
        if zero:
            t.zero_()
        elif self.poison and n:
            t.view(torch.uint8).fill_(0xFF)
        self.bufs.append((raw, m, n))
        return t.view(shape)

    def _wrap(self, zero):
        def fn(*size, dtype=None, device=None, **kw):
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            dev = torch.device(device) if device is not None else None
            if dev is None or dev.type != "cuda" or kw:
                return (self._zeros if zero else self._empty)(*size, dtype=dtype, device=device, **kw)
            return self.carve(shape, dtype or torch.float32, dev, zero)
        return fn

    def __enter__(self):
        torch.empty, torch.zeros = self._wrap(False), self._wrap(True)
        return self

    def __exit__(self, *a):
        torch.empty, torch.zeros = self._empty, self._zeros

    def check(self, what):
        torch.cuda.synchronize()
        for i, (raw, m, n) in enumerate(self.bufs):
            b = raw.view(torch.uint8)
            es = raw.element_size()
            head, tail = b[:m * es], b[(m + n) * es:]
            assert bool((head == PAT).all()), f"{what}: allocation {i} ({n} elements): bytes BEFORE it were written"
            assert bool((tail == PAT).all()), f"{what}: allocation {i} ({n} elements): bytes AFTER it were written"


@pytest.fixture()
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,H,L,hs,hd,causal,rel_hw,rope", [
    (2, 3, 77, 64, 64, False, None, False),
    (1, 3, 577, 64, 64, False, None, False),       # CLIP's shape: eight-wave forward / dQ, four-wave dK / dV
    (2, 2, 130, 96, 80, False, None, False),
    (1, 2, 1000, 96, 80, False, None, False),      # eight-wave forward, dQ, dK / dV; last key tile 40 rows
    (1, 2, 1024, 96, 80, False, (32, 32), False),  # SAM global: rel-pos instances
    (1, 3, 703, 128, 128, True, None, False),      # LLaMA: eight-wave forward, role-split dK / dV
    (2, 2, 703, 128, 128, True, None, True),       # + the fused inverse RoPE epilogues
    (1, 2, 200, 128, 128, True, None, True),
    (3, 2, 196, 96, 80, False, (14, 14), False),   # window kernels
])
def test_attention_kernels_write_inside_their_tensors(dev, B, H, L, hs, hd, causal, rel_hw, rope):
    from grove_amd import ops
    g = torch.Generator().manual_seed(5)
    alpha = hd ** -0.5
    with GuardedAllocs(poison=True) as ga:
        qkv = ga.carve((B * L, 3 * H * hs), bf16, dev, zero=True)
        qkv.view(B * L, 3, H, hs)[..., :hd] = torch.randn(B * L, 3, H, hd, generator=g).to(bf16).to(dev)
        do = ga.carve((B * L, H * hs), bf16, dev, zero=True)
        do.view(B * L, H, hs)[..., :hd] = torch.randn(B * L, H, hd, generator=g).to(bf16).to(dev)
        dqkv = ga.carve((B * L, 3 * H * hs), bf16, dev)
        rel, rel_arg = None, (0, 0)
        if rel_hw is not None:
            kh, kw = rel_hw
            khp = (kh + 15) // 16 * 16
            rel = ga.carve((B * H, L, khp + (kw + 15) // 16 * 16), bf16, dev, zero=True)
            rel[..., :kh] = (torch.randn(B * H, L, kh, generator=g) / alpha).to(bf16).to(dev)
            rel[..., khp:khp + kw] = (torch.randn(B * H, L, kw, generator=g) / alpha).to(bf16).to(dev)
            rel_arg = (khp, kw)
        hv = hd if hd < hs else 0
        out, lse = ops.flash_attn(qkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=causal, rel=rel, rel_hw=rel_arg, want_lse=True, hs_valid=hv)
        ga.check("forward")
        kw_b = {}
        if rope:
            kw_b["rope"] = ops.rope_table(hs, 10000.0, L + 5, dev)
        ops.flash_attn_bwd(qkv, out, do, lse, dqkv, B, L, H, hs, 0, H * hs, 2 * H * hs, alpha, causal=causal, rel=rel, rel_hw=rel_arg,
                           want_drel=rel is not None, hs_valid=hv, **kw_b)
        ga.check("backward")
    assert torch.isfinite(dqkv.float()).all() and torch.isfinite(out.float()).all()


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 7, 8])
@pytest.mark.parametrize("N,K,fold", [(520, 1096, False), (1000, 1024, False), (2064, 4096, True), (515, 11008, False), (16400, 256, False), (16400, 256, True)])
def test_gemv_kernels_write_inside_their_tensors(dev, M, N, K, fold):
    """VALU instances (K % 128 != 0 or M <= 2) and the matrix-core kernel (3..8 rows), plain and with the folded RMSNorm on an fp32 stream."""
    from grove_amd import ops
    g = torch.Generator().manual_seed(6)
    with GuardedAllocs(poison=True) as ga:
        w = ga.carve((N, K), bf16, dev)
        w.copy_((torch.randn(N, K, generator=g) * 0.05).to(bf16))
        x = ga.carve((M, K), torch.float32 if fold else bf16, dev)
        x.copy_(torch.randn(M, K, generator=g).to(x.dtype))
        res = ga.carve((M, N), torch.float32 if fold else bf16, dev)
        res.copy_(torch.randn(M, N, generator=g).to(res.dtype))
        nw = ga.carve((K,), bf16, dev)
        nw.copy_(torch.ones(K).to(bf16))
        y = ops.gemv(x, w, residual=res, rms_weight=nw if fold else None, eps=1e-5, out_dtype=res.dtype)
        ga.check(f"gemv M={M}")
    assert y.shape == (M, N) and torch.isfinite(y.float()).all()


@pytest.mark.parametrize("M,N,K,act", [
    (300, 200, 96, "gelu"), (2812, 520, 128, "none"), (130, 64, 64, "none"),
    (2812, 1288, 4096, "none"),     # persistent 192- / 256-row tiles with a ragged last row and column tile
    (4100, 1280, 1280, "gelu"),     # partial last round: stream-K parts + fix-up launch (workspace slots)
    (260, 4096, 8192, "none"),      # few tiles, long K: the only round cut into K ranges
    (703, 4096, 11008, "none"),
])
def test_gemm_kernels_write_inside_their_tensors(dev, M, N, K, act):
    """NT GEMM family: ragged edges, stream-K workspace, fix-up launch — C, the workspace and every other allocation keep their margins."""
    from grove_amd import ops
    g = torch.Generator().manual_seed(7)
    with GuardedAllocs(poison=True) as ga:
        a = ga.carve((M, K), bf16, dev)
        a.copy_(torch.randn(M, K, generator=g).to(bf16))
        b = ga.carve((N, K), bf16, dev)
        b.copy_((torch.randn(N, K, generator=g) * 0.05).to(bf16))
        bias = ga.carve((N,), bf16, dev)
        bias.copy_(torch.randn(N, generator=g).to(bf16))
        res = ga.carve((M, N), bf16, dev)
        res.copy_(torch.randn(M, N, generator=g).to(bf16))
        y = ops.linear(a, b, bias, act=ops.ACT_GELU if act == "gelu" else ops.ACT_NONE, residual=res)
        ga.check("nt gemm")
        gw = ga.carve((N, K), torch.float32, dev, zero=True)
        dy = ga.carve((M, N), bf16, dev)
        dy.copy_(torch.randn(M, N, generator=g).to(bf16))
        ops.wgrad(dy, a, gw)   # gw[N, K] += dy^T a
        ga.check("tn gemm")
    assert torch.isfinite(y.float()).all() and torch.isfinite(gw).all()


@pytest.mark.parametrize("rows,C,rms", [(703, 4096, True), (2812, 1024, True), (1000, 1280, False), (577, 1024, False), (37, 320, False)])
def test_norm_kernels_write_inside_their_tensors(dev, rows, C, rms):
    from grove_amd import ops
    g = torch.Generator().manual_seed(8)
    with GuardedAllocs(poison=True) as ga:
        x = ga.carve((rows, C), bf16, dev)
        x.copy_(torch.randn(rows, C, generator=g).to(bf16))
        w = ga.carve((C,), bf16, dev)
        w.copy_(torch.randn(C, generator=g).to(bf16))
        bvec = ga.carve((C,), bf16, dev, zero=True)
        dy = ga.carve((rows, C), bf16, dev)
        dy.copy_(torch.randn(rows, C, generator=g).to(bf16))
        if rms:
            y = ops.rmsnorm(x, w, 1e-5)
            dx = ops.rmsnorm_bwd(x, w, dy, 1e-5)
        else:
            y, mean, rstd = ops.layernorm(x, w, bvec, 1e-5, save_stats=True)
            dx = ops.layernorm_bwd(x, w, dy, mean, rstd)
        ga.check("norm")
    assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all()


@pytest.mark.parametrize("mode", ["train", "infer_generate", "infer_masks", "infer_fp8"])
def test_whole_model_writes_inside_its_tensors(dev, mode):
    """Every kernel of the path in one go: a tiny-dims training step (forward + backward: towers, window / global / causal attention,
    Conv3d adapters, decoder, losses) and an inference pass with cached greedy decoding at B = 3 (the padded-row GEMV instances), with
    every torch.empty / torch.zeros of the product code carved out of patterned buffers."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    d = TINY
    sd = synthetic_state_dict(d)
    train = mode == "train"
    extra = dict(gemm_dtype="fp8", fp8_policy="det16_kv16") if mode == "infer_fp8" else {}  # (CLIP in e4m3 too: every fp8 kernel of config 5)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, train=train, **extra)
    batch = synthetic_batch(d, B=2 if train else 3, T=8, L=48, n_det=2, seed=7, ragged=True)
    kw = batch.as_kwargs()
    for k in ("global_enc_images", "grounding_enc_images"):
        kw[k] = kw[k].to(dev).to(bf16)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kw[k] = kw[k].to(dev)
    if train:
        model.zero_grad()
    with GuardedAllocs(poison=True) as ga:
        if train:
            out = model(**kw)
            model.backward(out["loss"])
            ga.check("training step")
            assert torch.isfinite(out["loss"]).all() and torch.isfinite(model._flat_grad).all()
        elif mode == "infer_fp8":
            out = model(**dict(kw, inference=True))
            ga.check("fp8 inference forward")
            assert torch.isfinite(out["flat_boxes"]).all() and torch.isfinite(out["hidden"].float()).all()
        elif mode == "infer_masks":  # config 5's path: teacher-forced inference forward + the SAM mask branch with its post-processing
            out = model(**dict(kw, inference=True))
            n_inst = int(out["flat_boxes"].shape[0]) // 8
            text = torch.randn(max(n_inst, 1), d.out_dim, device=dev) * 0.1
            inst_frame = torch.zeros(max(n_inst, 1), dtype=torch.int32, device=dev)
            res = model.predict_masks(out["image_embeddings"], text, inst_frame, input_size=(384, 512), original_size=(360, 640))
            ga.check("inference forward + masks")
            assert torch.isfinite(out["flat_boxes"]).all() and torch.isfinite(res["masks"]).all() and torch.isfinite(res["iou_predictions"]).all()
        else:
            feats, _ = model(mode="encode_images", images=kw["global_enc_images"])
            prompt = kw["input_ids"][:, :20].contiguous()
            gen = model.generate(input_ids=prompt, image_features=feats, max_new_tokens=5, eos_token_id=-1, output_hidden_states=True,
                                 return_dict_in_generate=True)
            ga.check("generate B=3")
            assert gen.sequences.shape[0] == 3
            assert all(torch.isfinite(h.float()).all() for h in gen.hidden_states), "a generated hidden state read an element nobody wrote"
    assert len(ga.bufs) > 50


def test_full_dims_bench_step_writes_inside_its_tensors(dev):
    """The headline step itself (full dims, B = 2 clips x T = 16: the persistent GEMMs' stream-K cuts and fix-up slots at M = 2812 / 32768,
    temporal tap skipping, the last-layer tail, the eight-wave attention kernels on their real shapes, window kernels with real-token
    lists) under the same screen: the tiny-dims step above takes other tile plans."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    d = FULL
    sd = synthetic_state_dict(d, device=dev, dtype=bf16)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, train=True)
    del sd
    torch.cuda.empty_cache()
    kw = synthetic_batch(d, B=2, T=16, L=128, n_det=3, seed=11).as_kwargs()
    for k in ("global_enc_images", "grounding_enc_images"):
        kw[k] = kw[k].to(dev).to(bf16)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kw[k] = kw[k].to(dev)
    model.zero_grad()
    out = model(**kw)                      # warm: plans, workspaces, tables
    model.backward(out["loss"])
    model.zero_grad()
    with GuardedAllocs(poison=True) as ga:
        out = model(**kw)
        model.backward(out["loss"])
        ga.check("full-dims training step")
    assert torch.isfinite(out["loss"]).all() and len(ga.bufs) > 200
