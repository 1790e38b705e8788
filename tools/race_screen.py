"""Race screen for the pipelined kernels (counted vmcnt + raw barriers: a misplaced wait passes single runs): many launches per
shape, each compared bit for bit with the two-barrier kernels' result on the same operands, while another stream keeps the
memory system busy. Shapes cover 1-3 K tiles (prologue / tail paths), edge tiles, epilogue variants, gathers, TN."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops, _lib
from grove_amd.model.indexing import conv3d_gather_index
dev = torch.device("cuda:0")
L = _lib.lib()
bf = torch.bfloat16
torch.manual_seed(0)
noise_a = torch.randn(64 << 20, device=dev)
side = torch.cuda.Stream()
bad = 0
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def churn():
    with torch.cuda.stream(side):
        noise_a.mul_(1.0001)


def screen(name, fn_ref, fn_new):
    global bad
    ref = fn_ref()
    ref = [r.clone() for r in (ref if isinstance(ref, tuple) else (ref,))]
    fails = 0
    for i in range(REPS):
        if i % 3 == 0:
            churn()
        out = fn_new()
        out = out if isinstance(out, tuple) else (out,)
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            fails += 1
    torch.cuda.synchronize()
    print(f"{name:55s} {'OK' if fails == 0 else 'MISMATCH x%d' % fails}", flush=True)
    bad += fails


L.grove_gemm_set_stream_k(0)  # bit-for-bit against the two-barrier kernels needs whole tiles (a split K sum rounds differently); the split is screened below
for (M, N, K) in [(2812, 4096, 64), (2812, 4096, 128), (2812, 4096, 192), (3000, 520, 1280), (32768, 1280, 256), (2812, 12288, 4096), (8200, 5120, 320)]:
    a = torch.randn(M, K, device=dev).to(bf); b = (torch.randn(N, K, device=dev) * 0.1).to(bf)
    bias = torch.randn(N, device=dev).to(bf); res = torch.randn(M, N, device=dev).to(bf)
    for tm in (256, 193):
        for kw_name, kw in (("bias", dict(bias=bias)), ("gelu+aux", dict(bias=bias, act=ops.ACT_GELU, aux=True)), ("residual", dict(bias=bias, residual=res))):
            def run(tile):
                L.grove_gemm_set_tile_m(tile)
                k2 = dict(kw)
                aux = None
                if k2.pop("aux", False):
                    aux = torch.empty(M, N, device=dev, dtype=bf)
                    k2["aux"] = aux
                o = ops.linear(a, b, **k2)
                return (o, aux) if aux is not None else o
            screen(f"NT {M}x{N}x{K} tile {tm} {kw_name}", lambda: run(128), lambda: run(tm))
L.grove_gemm_set_tile_m(0)
# stream-K tail (parts through the workspace + fix-up launch): deterministic by construction — screened against its own first result,
# whole-tile result within bf16 rounding
L.grove_gemm_set_stream_k(1)
for (M, N, K, tm) in [(32768, 1280, 5120, 256), (2812, 12288, 4096, 256), (20200, 1000, 2560, 256), (15350, 1000, 2560, 193)]:
    a = torch.randn(M, K, device=dev).to(bf); b = (torch.randn(N, K, device=dev) * 0.1).to(bf)
    bias = torch.randn(N, device=dev).to(bf); res = torch.randn(M, N, device=dev).to(bf)
    def sk():
        L.grove_gemm_set_tile_m(tm)
        return ops.linear(a, b, bias, residual=res)
    first = sk()
    assert L.grove_gemm_last_stream_k() > 0, (M, N, K, tm)
    L.grove_gemm_set_stream_k(0)
    whole = sk()
    L.grove_gemm_set_stream_k(1)
    assert (first.float() - whole.float()).abs().max().item() <= 2 ** -7 * whole.float().abs().max().item()
    screen(f"NT {M}x{N}x{K} tile {tm} stream-K (self)", sk, sk)
L.grove_gemm_set_stream_k(0)
L.grove_gemm_set_tile_m(0)
# gathered A (27 taps) and SwiGLU pair
G, T, H, W, Ci, Co = 2, 8, 16, 16, 128, 512
Mt = G * T * H * W
x = torch.randn(Mt, Ci, device=dev).to(bf); w = (torch.randn(Co, 27 * Ci, device=dev) * 0.05).to(bf)
idx = conv3d_gather_index(G, T, H, W).to(dev)
def conv(tile):
    L.grove_gemm_set_tile_m(tile)
    return ops.linear(x, w, a_idx=idx, a_taps=27, M=Mt)
screen("NT gathered conv3d tile 256", lambda: conv(128), lambda: conv(256))
screen("NT gathered conv3d tile 192", lambda: conv(128), lambda: conv(193))
L.grove_gemm_set_tile_m(0)
xx = torch.randn(2812, 512, device=dev).to(bf); wgu = (torch.randn(2 * 2752, 512, device=dev) * 0.1).to(bf); wsw = ops.swiglu_interleave(wgu)
screen("NT swiglu pair", lambda: ops.swiglu(ops.linear(xx, wgu), 2752), lambda: ops.linear(xx, wsw, act=ops.ACT_SWIGLU_PAIR))
# TN pipelined (plain + gathered)
for (K2, M2, N2) in [(64, 520, 776), (192, 1280, 2560), (4096, 1000, 1032)]:
    dy = torch.randn(K2, M2, device=dev).to(bf); xb = torch.randn(K2, N2, device=dev).to(bf)
    def tn(mode):
        L.grove_gemm_tn_set_pipelined(mode)
        return ops.wgrad(dy, xb, torch.zeros(M2, N2, dtype=torch.float32, device=dev))
    r0, r1 = tn(0), tn(1)
    assert (r0 - r1).abs().max().item() <= 2e-6 * r0.abs().max().item(), "TN pipelined vs 128x128 kernel"
    # the 128 x 128 kernel splits K with atomics on these shapes (order-dependent fp32 sums): the pipelined kernel is screened
    # against its own first result
    screen(f"TN {K2}x{M2}x{N2} (self)", lambda: tn(1), lambda: tn(1))
dz = torch.randn(Mt, 264, device=dev).to(bf); x2 = torch.randn(Mt, 256, device=dev).to(bf)
def tng(mode):
    L.grove_gemm_tn_set_pipelined(mode)
    return ops.wgrad(dz, x2, torch.zeros(264, 27 * 256, dtype=torch.float32, device=dev), b_idx=idx, b_taps=27)
r0, r1 = tng(0), tng(1)
assert (r0 - r1).abs().max().item() <= 2e-6 * r0.abs().max().item(), "TN gathered pipelined vs 128x128 kernel"
screen("TN gathered conv3d wgrad (self)", lambda: tng(1), lambda: tng(1))
L.grove_gemm_tn_set_pipelined(-1)
L.grove_gemm_set_stream_k(1)
# ---- round 4 ----
# temporal tap skipping (short tiles dealt after the full ones: another work list, another vmcnt tail per block): whole tiles are
# bit-identical to the un-skipped launch, which is the reference here
G4, T4, H4, W4, Ci4, Co4 = 4, 8, 32, 32, 128, 256
M4 = G4 * T4 * H4 * W4
x4 = torch.randn(M4, Ci4, device=dev).to(bf); w4 = (torch.randn(Co4, 27 * Ci4, device=dev) * 0.03).to(bf); b4 = torch.randn(Co4, device=dev).to(bf)
idx4 = conv3d_gather_index(G4, T4, H4, W4).to(dev)
L.grove_gemm_set_stream_k(0)
def conv4(skip):
    L.grove_gemm_set_tap_skip(skip)
    return ops.linear(x4, w4, b4, a_idx=idx4, a_taps=27, M=M4, a_frames=(H4 * W4, T4))
screen("NT conv3d tap skip vs all taps", lambda: conv4(0), lambda: conv4(1))
L.grove_gemm_set_stream_k(1)
L.grove_gemm_set_tap_skip(1)
dz4 = torch.randn(M4, 512, device=dev).to(bf); xx4 = torch.randn(M4, 256, device=dev).to(bf)
L.grove_gemm_tn_set_pipelined(1); L.grove_gemm_tn_set_split_tail(0)
def tn4(skip):
    L.grove_gemm_tn_set_tap_skip(skip)
    return ops.wgrad(dz4, xx4, torch.zeros(512, 27 * 256, dtype=torch.float32, device=dev), b_idx=idx4, b_taps=27, b_frames=(H4 * W4, T4))
screen("TN conv3d wgrad tap skip vs all taps", lambda: tn4(0), lambda: tn4(1))
L.grove_gemm_tn_set_pipelined(-1); L.grove_gemm_tn_set_split_tail(1); L.grove_gemm_tn_set_tap_skip(1)
# attention backward with the inverse RoPE in its epilogues, and the tail form (Lq != Lk): self-consistency under churn
Bq, Hq, hq, Sq = 2, 8, 128, 703
qkv = (torch.randn(Bq * Sq, 3 * Hq * hq, device=dev) * 0.5).to(bf)
do_ = (torch.randn(Bq * Sq, Hq * hq, device=dev) * 0.5).to(bf)
o_, lse_ = ops.flash_attn(qkv, Bq, Sq, Hq, hq, 0, Hq * hq, 2 * Hq * hq, hq ** -0.5, causal=True, want_lse=True)
tab = ops.rope_table(hq, 10000.0, 1024, dev)
def fbwd():
    dqkv = torch.empty_like(qkv)
    ops.flash_attn_bwd(qkv, o_, do_, lse_, dqkv, Bq, Sq, Hq, hq, 0, Hq * hq, 2 * Hq * hq, hq ** -0.5, causal=True, rope=tab)
    return dqkv
screen("flash bwd + fused inverse RoPE (self)", fbwd, fbwd)
Lq4 = 54
q_t = (torch.randn(Bq * Lq4, Hq * hq, device=dev) * 0.5).to(bf); kv_ = qkv[:, Hq * hq:].contiguous()
ot, lset = ops.flash_attn_tail(q_t, kv_, Bq, Lq4, Sq, Hq, hq, hq ** -0.5, want_lse=True)
dot_ = (torch.randn(Bq * Lq4, Hq * hq, device=dev) * 0.5).to(bf)
def tbwd():
    dq, dkv = torch.empty_like(q_t), torch.empty_like(kv_)
    ops.flash_attn_tail_bwd(q_t, kv_, ot, dot_, lset, dq, dkv, Bq, Lq4, Sq, Hq, hq, hq ** -0.5, rope=tab)
    return dq, dkv
screen("tail attention bwd + fused inverse RoPE (self)", tbwd, tbwd)
# round 5: the eight-wave attention kernels (LDS-DMA rings, counted waits, two wave halves half a step apart) on the step's three
# shapes, forward and backward, repeated under churn: self-consistency bit for bit, and the forward against the four-wave kernels'
# result within bf16 rounding of the different summation order
def attn_case(name, B, H, Lc, hs, hd, causal, rel_hw):
    g = torch.Generator(device="cpu").manual_seed(3)
    q3 = torch.zeros(B * Lc, 3, H, hs)
    q3[..., :hd] = torch.randn(B * Lc, 3, H, hd, generator=g) * 0.7
    qkv_c = q3.view(B * Lc, 3 * H * hs).to(bf).to(dev)
    d3 = torch.zeros(B * Lc, H, hs)
    d3[..., :hd] = torch.randn(B * Lc, H, hd, generator=g)
    do_c = d3.view(B * Lc, H * hs).to(bf).to(dev)
    al = hd ** -0.5
    rel_c, ra = None, (0, 0)
    if rel_hw:
        kh, kw = rel_hw
        rel_c = torch.zeros(B * H, Lc, kh + kw)
        rel_c[..., :kh + kw] = torch.randn(B * H, Lc, kh + kw, generator=g) / al
        rel_c, ra = rel_c.to(bf).to(dev), (kh, kw)
    hv = hd if hd < hs else 0
    def fwd():
        return ops.flash_attn(qkv_c, B, Lc, H, hs, 0, H * hs, 2 * H * hs, al, causal=causal, rel=rel_c, rel_hw=ra, want_lse=True, hs_valid=hv)
    screen(name + " eight-wave fwd (self)", fwd, fwd)
    o_c, lse_c = fwd()
    def bwd():
        dq_c = torch.zeros_like(qkv_c)
        dr = ops.flash_attn_bwd(qkv_c, o_c, do_c, lse_c, dq_c, B, Lc, H, hs, 0, H * hs, 2 * H * hs, al, causal=causal, rel=rel_c, rel_hw=ra,
                                want_drel=rel_c is not None, hs_valid=hv, rope=(tab if causal else None))
        return (dq_c, dr) if dr is not None else dq_c
    screen(name + " eight-wave bwd (self)", bwd, bwd)
    L.grove_flash_attn_set_v2(0)
    o4, _ = fwd()
    L.grove_flash_attn_set_v2(47)
    err = (o_c.float() - o4.float()).abs().max().item()
    print(f"{name + ' eight-wave vs four-wave forward':55s} max |diff| {err:.4g}", flush=True)
    assert err < 2e-2
attn_case("SAM global <96, rel>", 4, 16, 1024, 96, 80, False, (32, 32))
attn_case("LLaMA <128> causal", 4, 32, 703, 128, 128, True, None)
attn_case("CLIP <64>", 8, 16, 577, 64, 64, False, None)
# the whole training step in deterministic mode (every overlap on): repeated steps from the same state must agree bit for bit
ops.set_deterministic(True)
from grove_amd import train as TR
from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
def three_steps():
    args = TR.shipped_args(); args.lr, args.steps_per_epoch, args.print_freq = 1e-3, 4, 2
    eng = TR.GroveEngine(TR.initialize_model(args, dims=TINY, state_dict=synthetic_state_dict(TINY), device=dev), args, total_steps=1000)
    eng.scheduler.warm = 0
    ls = []
    for st in range(3):
        kw = synthetic_batch(TINY, B=2, T=8, L=48, n_det=2, seed=10 + st, ragged=True).as_kwargs()
        for k in ("global_enc_images", "grounding_enc_images"):
            kw[k] = kw[k].to(dev).to(bf)
        for k in ("input_ids", "labels", "attention_masks", "offset"):
            kw[k] = kw[k].to(dev)
        out = eng(**kw); ls.append(out["loss"]); eng.backward(out["loss"]); eng.step()
    torch.cuda.synchronize()
    return torch.stack(ls), eng.master.clone()
REPS = max(REPS // 6, 4)
screen("3 training steps, deterministic mode (self)", three_steps, three_steps)
# ... and at FULL size (the shapes, grids and stream overlaps the bench runs): one engine, the same batch three times from the same
# weights (gradients and losses compared; no optimizer step), towers overlapped
if "--no-full" not in sys.argv:
    from grove_amd.synthetic import FULL
    args = TR.shipped_args(); args.lr, args.num_frames, args.batch_size = 1e-4, 16, 2
    eng = TR.GroveEngine(TR.initialize_model(args, dims=FULL, state_dict=synthetic_state_dict(FULL, device=dev, dtype=bf), device=dev), args, total_steps=1000)
    kwf = synthetic_batch(FULL, B=2, T=16, L=128, n_det=3, seed=3, device=dev, dtype=bf).as_kwargs()
    def full_step():
        eng.module.zero_grad()
        out = eng(**kwf)
        eng.backward(out["loss"])
        torch.cuda.synchronize()
        return torch.stack([out[k].detach().float().reshape(()) for k in sorted(out) if k.endswith("loss")]), eng.module._flat_grad.clone()
    REPS = 3
    screen("full-size fwd + bwd (2 clips x T = 16), deterministic mode (self)", full_step, full_step)
ops.set_deterministic(False)
# ---- round 6: the Winograd pipeline's two GEMM modes, its transforms, and the request-shape loads of the matrix-core GEMV
L.grove_gemm_set_stream_k(0)
L.grove_gemm_set_tile_m(256)
for groups, rows, N, K in ((64, 512, 1280, 1280), (12, 256, 320, 192)):
    ga = torch.randn(groups * rows, K, device=dev).to(bf); gb = (torch.randn(groups, N, K, device=dev) * 0.1).to(bf)

    def grouped():
        o = torch.empty(groups * rows, N, device=dev, dtype=bf)
        ops.gemm_raw(ga, gb, o, groups * rows, N, K, K, K, N, b_group=rows)
        return o

    def per_group():
        return torch.cat([ops.linear(ga[g_ * rows:(g_ + 1) * rows], gb[g_]) for g_ in range(groups)], 0)
    screen(f"NT grouped B {groups} x [{rows}, {K}] x [{N}, {K}] vs per-group launches", per_group, grouped)
L.grove_gemm_set_tile_m(0)
L.grove_gemm_set_stream_k(1)
L.grove_gemm_tn_set_pipelined(1)
L.grove_gemm_tn_set_split_tail(0)
for batches, K, M, N in ((64, 1024, 1280, 1280), (7, 192, 264, 136)):
    ta = torch.randn(batches * K, M, device=dev).to(bf); tb = torch.randn(batches * K, N, device=dev).to(bf)

    def kbatched():
        o = torch.empty(batches, M, N, device=dev)
        ops.wgrad(ta, tb, o, K=K, k_batches=batches, sC_batch=M * N, overwrite=True, M=M, N=N)
        return o

    def per_batch():
        o = torch.zeros(batches, M, N, device=dev)
        for b_ in range(batches):
            ops.wgrad(ta[b_ * K:(b_ + 1) * K], tb[b_ * K:(b_ + 1) * K], o[b_])
        return o
    screen(f"TN K-batched {batches} x [{K}, {M}]^T [{K}, {N}] vs per-batch launches", per_batch, kbatched)
L.grove_gemm_tn_set_pipelined(-1)
L.grove_gemm_tn_set_split_tail(1)
wgeom, wC = (2, 8, 16, 16), 320
wx = torch.randn(wgeom[0] * wgeom[1] * wgeom[2] * wgeom[3], wC, device=dev).to(bf)
ww = (torch.randn(wC, 27 * wC, device=dev) * 0.05).to(bf)
wa = torch.tensor([0.2], device=dev)


def wino_all():
    y, pre = torch.empty_like(wx), torch.empty_like(wx)
    _, V = ops.wino3d_conv(wx, ops.wino3d_transform_weight(ww), wgeom, y, act=ops.ACT_RELU, scale_ptr=wa, scale_tanh=True, residual=wx, aux=pre, keep_V=True)
    gw = torch.zeros(wC, 27 * wC, device=dev)
    ops.wino3d_wgrad(pre, V, wgeom, gw, scale_ptr=wa, scale_tanh=True)
    return y, pre, gw
screen("Winograd adapter forward + wgrad (self)", wino_all, wino_all)
for N, K in ((4096, 4096), (22016, 4096), (4096, 11008)):
    vw = (torch.randn(N, K, device=dev) * 0.02).to(bf); vx = torch.randn(8, K, device=dev).to(bf)

    def gv(knob):
        L.grove_gemv_set_mfma(knob)
        o = ops.gemv(vx, vw)
        L.grove_gemv_set_mfma(1)
        return o
    screen(f"GEMV 8 x [{N}, {K}]: 128 B / row loads + lane exchange vs operand-shape loads", lambda: gv(9), lambda: gv(1))
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
