"""Headline benchmark: frames/sec of the GROVE training step (T=16 clips, per-GPU batch 2, bf16, fwd+bwd+
gradient exchange+AdamW) on N MI355X of one node — BASELINE.json `metric`, config[2] ("train.py iGround
fine-tune, T=16, per-GPU batch=2, bf16"), config[3] for N>1.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0. `value` = total frames/s over all ranks with inputs resident in HBM; `roofline`
prices the dominant kernel (the persistent pipelined bf16 MFMA GEMM) from HIP-event timings of its own launches in one
extra, untimed, instrumented step with the two towers serialised (in the timed steps the SAM tower runs on a second stream beside
CLIP -> LLaMA, so two kernels share the CUs and a launch's wall time is not the kernel's own; `--serial_towers` runs everything that
way and is the command the committed rocprofv3 summaries come from; the other GEMM kernels are summarised beside it); `cpu_baseline` times the CPU oracle (a port, oracle/grove_oracle.py) on a bounded
sample of the same workload on the host cores. Synthetic data and random-init weights of the real
architecture (no checkpoints offline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP8_BOX_L1_BOUND = 9.5e-3  # e4m3's own figure on the tiny case (policy det16_kv16: 6.3e-3 measured, 6.1e-3 predicted by tools/fp8_policy_study.py) x 1.5; the default policy since round 4 (+ CLIP in bf16) sits below it
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)


def build(dims, dev, args):
    from grove_amd import train as T
    from grove_amd.synthetic import synthetic_state_dict
    targs = T.shipped_args()  # --lora_r 0 --pretrained --train_mask_decoder: every shipped launch line (train_scripts/*.sh)
    targs.num_frames = args.frames
    targs.batch_size = args.batch
    if getattr(args, "stream", "default") != "default":
        targs.stream_dtype = args.stream
    sd = synthetic_state_dict(dims, device=dev, dtype=torch.bfloat16)
    if getattr(args, "clip_alpha", 0.0):  # a TRAINED checkpoint's CLIP adapters are active (SURVEY's synthetic weights: alpha = 0, the conv is skipped)
        for k in sd:
            if "vision_tower" in k and k.endswith(".alpha"):
                sd[k] = torch.full_like(sd[k], args.clip_alpha)
    model = T.initialize_model(targs, dims=dims, state_dict=sd, device=dev)
    del sd
    torch.cuda.empty_cache()
    engine = T.GroveEngine(model, targs, total_steps=100000, exchange=(lambda e: "allreduce" if e == "auto" else e)(getattr(args, "exchange", "allreduce")),
                           overlap=not getattr(args, "no_comm_overlap", False), sparse_embed=not getattr(args, "dense_embed", False))
    return model, engine


def make_batch(dims, dev, args, rank):
    from grove_amd.synthetic import synthetic_batch
    b = synthetic_batch(dims, B=args.batch, T=args.frames, L=args.text_len, n_det=3, seed=1000 * rank + 7, device=dev,
                        dtype=torch.bfloat16)
    return b.as_kwargs(inference=False)


def instrumented_gemm_pass(engine, batch):
    """One extra step with a HIP-event pair around every grove_gemm_bf16 launch (torch's current stream is the
    stream the kernels are launched on). Returns per-kernel {variant: (launches, algorithmic flops, device seconds)}
    keyed by the kernel the library actually launched (grove_gemm_last_variant)."""
    from grove_amd import _lib, ops
    orig = ops.gemm_raw
    recs = []
    names = {1: "gemm_nt_kernel<128x128>", 2: "gemm_nt_kernel<192x128>", 3: "gemm_nt_kernel<128x64>"}
    pp = {4: "256, false", 5: "192, false", 6: "256, true", 7: "192, true"}  # + the epilogue: the instance name rocprofv3 prints

    def kernel_name(grouped=False):
        v = _lib.lib().grove_gemm_last_variant()
        if v in pp:
            return f"gemm_nt_pp_kernel<{pp[v]}, {_lib.lib().grove_gemm_last_epilogue()}" + (", false, true>" if grouped else ">")
        return names.get(v, "?")

    def timed(A, B, C, M, N, K, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(A, B, C, M, N, K, *a, **k)
        e1.record()
        b = k.get("batch", (1, 1))
        grouped = bool(k.get("b_group", 0))
        recs.append((e0, e1, 2.0 * M * N * K * b[0] * b[1], (M, N, K, b[0] * b[1], k.get("a_taps", 1)),
                     kernel_name(grouped)))
        if grouped:
            # the Winograd form of a 3x3x3 Conv3d: M = 64 transform points x tiles of 8 outputs; the convolution it stands for is
            # 2 * (8 * tiles) * N * 27 * K flop (2 * MAC of the direct form) — the step figure counts THAT
            instrumented_gemm_pass.algorithmic_extra += 2.0 * (M // 8) * N * 27 * K - 2.0 * M * N * K
        return r
    ops.gemm_raw = timed
    instrumented_gemm_pass.algorithmic_extra = 0.0
    overlap = engine.module.tower_overlap
    engine.module.tower_overlap = False  # a kernel is priced on its own: with the SAM tower on a second stream two kernels share the CUs
    try:
        out = engine(**batch)
        engine.backward(out["loss"])
        engine.step()
        torch.cuda.synchronize()
    finally:
        ops.gemm_raw = orig
        engine.module.tower_overlap = overlap
    per_kernel = {}
    for e0, e1, f, key, var in recs:
        n, fl, t = per_kernel.get(var, (0, 0.0, 0.0))
        per_kernel[var] = (n + 1, fl + f, t + e0.elapsed_time(e1) * 1e-3)
    report = os.environ.get("GROVE_GEMM_REPORT")
    if report:
        agg = {}
        for e0, e1, f, key, var in recs:
            t, n, fl = agg.get(key + (var,), (0.0, 0, 0.0))
            agg[key + (var,)] = (t + e0.elapsed_time(e1), n + 1, fl + f)
        with open(report, "w") as fh:
            fh.write("M N K batch taps kernel | launches total_ms TF/s\n")
            for key, (t, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                fh.write(f"{key} | {n} {t:.3f} {fl / t / 1e9:.1f}\n")
    return per_kernel


def vit_llama_forward(model, batch, dims, reps=5):
    """north_star's stage figure: the fused CLIP-ViT + projector + LLaMA FORWARD at the bench batch (no SAM, no heads), timed with
    HIP events; algorithmic FLOPs per SURVEY.md section 8(d): per frame CLIP 0.69 + 23 x (2*577*12.58e6 + 4*577^2*1024) flop
    (layer 24's output is never read), 24 GF per 8-frame group for the projector, per sequence S*2*6.476e9 + 2*S^2*4096*32."""
    from grove_amd.model.GROVE import bf
    d = dims
    g = model._windows(batch["global_enc_images"], batch["grounding_enc_images"], batch["input_ids"], None, None, [None, None])
    gimg, ids = g[0], g[2]
    B = ids.shape[0]

    def run():
        feats, _ = model.encode_images(gimg)
        plan = model._splice_plan(ids, None, None, list(range(B)))
        x = model._embed(plan, feats.data if hasattr(feats, "data") else feats)
        hidden, _ = model.llama.forward(x, plan.B, plan.S)
        return plan.S
    with torch.no_grad():
        S = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    frames = gimg.shape[0] * gimg.shape[2]
    tok = 577
    clip = frames * (0.69e9 + (d.clip_layers - 1) * (2 * tok * 12.58e6 + 4 * tok * tok * d.clip_dim))
    proj = (frames // 8) * 24e9
    llama = B * (S * 2 * 6.476e9 + 2.0 * S * S * d.hidden * d.n_layers)
    flops = clip + proj + llama
    return {"ms": round(ms, 2), "flops": flops, "achieved": round(flops / ms / 1e9, 1), "unit": "TFLOP/s",
            "frac_of_bf16_peak": round(flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4), "windows": B, "S": S}


def decode_rate(model, dims, dev):
    """Cached greedy decode on the bench's own model (config 2's inner loop): ms per generated token at B = 1 after a 615-position
    prefill, and the fraction of the HBM peak the weight stream reaches (13.2 GB of bf16 weights are read once per token)."""
    from grove_amd.synthetic import synthetic_batch
    b = synthetic_batch(dims, B=1, T=8, L=64, n_det=1, seed=5, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        feats, _ = model.encode_images(b.global_enc_images)
        prompt = b.input_ids[:, :40].contiguous()

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(input_ids=prompt, image_features=feats, max_new_tokens=n, eos_token_id=-1)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        run(3)
        # (every call pays its own prefill + graph capture, ~35 ms with a few ms of jitter: the difference of two lengths cancels
        # them, the minimum of two runs each and a 48-step span keep the jitter out of the per-token figure)
        t9, t57 = min(run(9), run(9)), min(run(57), run(57))
    per_tok = (t57 - t9) / 48
    wbytes = 2.0 * (dims.n_layers * (4 * dims.hidden * dims.hidden + 3 * dims.hidden * dims.mlp) + dims.vocab * dims.hidden)
    return {"ms_per_token": round(per_tok * 1e3, 3), "tokens_per_s": round(1.0 / per_tok, 1), "batch": 1, "prefill_positions": 575 + 40,
            "roofline": {"bound": "hbm", "achieved": round(wbytes / per_tok / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(wbytes / per_tok / 8e12, 3), "weight_bytes_per_token": wbytes}}


def cpu_layer_samples(args):
    """Per-layer samples of the oracle at full dims (one layer of each tower, forward and — where the step has one — backward):
    the backward part of the CPU baseline, and the whole of it in the `--cpu_baseline sampled` fallback."""
    from grove_amd.synthetic import FULL, det_tensor, param_shapes, synthetic_state_dict
    from oracle import grove_oracle as O
    d = FULL
    want = [n for n in param_shapes(d) if any(s in n for s in (
        "vision_model.encoder.layers.1.", "image_encoder.blocks.6.", "image_encoder.blocks.7.", "image_encoder.adapters.0.",
        "model.layers.0."))]
    sd = synthetic_state_dict(d, names=set(want))

    def timed(fn, rep=1):
        t0 = time.perf_counter()
        for _ in range(rep):
            fn()
        return (time.perf_counter() - t0) / rep
    x = det_tensor("cpu.clip", (8, 577, d.clip_dim), std=1.0)
    with torch.no_grad():
        t_clip = timed(lambda: O.clip_layer(sd, d, 1, x))
    xs = det_tensor("cpu.sam", (8, 32, 32, d.sam_dim), std=1.0)

    def sam_fb(i):
        xi = xs.clone().requires_grad_(True)
        t0 = time.perf_counter()
        y = O.sam_block(sd, d, i, xi)
        t1 = time.perf_counter()
        y.sum().backward()
        return t1 - t0, time.perf_counter() - t1
    swf, swb = sam_fb(6)
    sgf, sgb = sam_fb(7)
    for n in want:
        if "adapters.0" in n:
            sd[n].requires_grad_(True)
    xa = xs.clone().requires_grad_(True)
    t0 = time.perf_counter()
    ya = O.sam_adapter(sd, d, 0, xa)
    t1 = time.perf_counter()
    ya.sum().backward()
    saf, sab = t1 - t0, time.perf_counter() - t1
    S = 575 + args.text_len
    xl = det_tensor("cpu.llama", (1, S, d.hidden), std=1.0).requires_grad_(True)
    cos, sin = O._rope_cos_sin(d, torch.arange(S))
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None]
    t0 = time.perf_counter()
    yl = O.llama_layer(sd, d, 0, xl, cos, sin, mask)
    t1 = time.perf_counter()
    yl.sum().backward()
    lf, lb = t1 - t0, time.perf_counter() - t1
    n_glob = len(d.sam_global)
    n_bwd = d.sam_depth - (min(d.sam_global) + 1)
    n_bwd_glob = n_glob - 1
    fwd = t_clip * (d.clip_layers - 1) + swf * (d.sam_depth - n_glob) + sgf * n_glob + saf * n_glob + lf * d.n_layers
    bwd = swb * (n_bwd - n_bwd_glob) + sgb * n_bwd_glob + sab * n_glob + lb * d.n_layers
    measured = t_clip + swf + swb + sgf + sgb + saf + sab + lf + lb
    return fwd, bwd, measured, S


def cpu_config1_leg(sd, d, fwd_fp32_s, cores, new_tokens=64, budget_s=150.0):
    """SURVEY.md section 8(d)'s config-1 CPU baseline (BASELINE config 1: one clip x 8 frames @336 px, greedy decode — the reference's own
    CPU-runnable case, grove_transformers / infer_iground.py:185-196): B = 1, T = 8 forward + `new_tokens` greedy tokens, on the oracle
    with its own K / V cache (oracle.llama_forward_cached). fp32: the window forward measured by the caller (the same towers + prefill) +
    the decode loop timed here. bf16: the LLaMA prefill + decode loop with bf16 weights and activations (torch CPU bf16 kernels) — the
    decode-dominated part; the towers are reported in fp32 only. Bounded: a leg that would pass `budget_s` stops early and is scaled."""
    import torch.nn.functional as Fn
    from grove_amd.synthetic import synthetic_batch
    from oracle import grove_oracle as O
    batch = synthetic_batch(d, B=1, T=8, L=64, n_det=1, seed=5)
    prompt = batch.input_ids[:, :40].contiguous()
    out = {"what": f"B=1, T=8 forward + {new_tokens}-token greedy decode (KV-cached oracle), {cores} threads", "new_tokens": new_tokens}
    t_all = time.perf_counter()
    with torch.no_grad():
        feats = torch.zeros(1, 576, d.hidden)  # (timing only: the visual tokens' values do not change the work)
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            # the LLaMA side of the state dict in the leg's dtype, converted ONCE, outside the timed regions (13.5 GB more for bf16)
            v = {k: t.detach().to(dt) for k, t in sd.items() if k.startswith("model.layers.") or k in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")}
            emb_table, lm_head = v["model.embed_tokens.weight"], v["lm_head.weight"]
            cache = []
            t0 = time.perf_counter()
            embeds, _, _ = O.splice(v, prompt, None, None, feats.to(emb_table.dtype))
            hidden = O.llama_forward_cached(v, d, embeds, cache)
            t_prefill = time.perf_counter() - t0
            t0 = time.perf_counter()
            done = 0
            for _ in range(new_tokens):
                tok = Fn.linear(hidden[:, -1], lm_head).argmax(-1)
                hidden = O.llama_forward_cached(v, d, emb_table[tok][:, None], cache)
                done += 1
                if time.perf_counter() - t_all > budget_s * (0.5 if name == "fp32" else 1.0):
                    break
            t_dec = (time.perf_counter() - t0) * new_tokens / max(done, 1)
            out[name] = {"llama_prefill_s": round(t_prefill, 2), "decode_s": round(t_dec, 2), "decode_tokens_measured": done,
                         "s_per_token": round(t_dec / new_tokens, 3)}
    total32 = fwd_fp32_s + out["fp32"]["decode_s"]
    out["fp32"]["forward_window_s"] = round(fwd_fp32_s, 1)
    out["fp32"]["clip_seconds"] = round(total32, 1)
    out["fp32"]["frames_per_s"] = round(8.0 / total32, 4)
    return out


def cpu_baseline(args, dev=None):
    """The CPU oracle (a port, oracle/grove_oracle.py) on the host cores, on a bounded sample of the bench workload: ONE of the
    step's 8-frame windows at FULL dimensions.
      default (`--cpu_baseline window`): the whole window FORWARD — SAM tower, CLIP tower, projector, splice, LLaMA 32 layers, box
        decoder — is run and timed end to end (about a minute on 64 threads; weights are regenerated tensor by tensor, their
        generation time excluded), and the step's backward is added from per-layer fwd+bwd samples scaled by the layer counts;
      `--cpu_baseline train`: the whole window fwd + bwd through torch autograd, timed (needs ~80 GB of host memory);
      `--cpu_baseline sampled`: per-layer samples only (the round-1/2 figure; a few seconds)."""
    from grove_amd.synthetic import FULL, synthetic_batch
    from oracle import grove_oracle as O
    d = FULL
    cores = min(os.cpu_count() or 1, 64)  # torch CPU kernels stop scaling (and oversubscribe) beyond ~64 threads
    torch.set_num_threads(cores)
    mode = args.cpu_baseline
    if mode == "auto":  # the whole fwd+bwd MEASURED when the host can hold it (~80 GB: fp32 weights + autograd's saved activations)
        mode = "train" if host_memory_available_gb() >= 128 else "window"
    cpu = cpu_model_name()
    if mode == "train":
        fwd_s = bwd_s = sampled = 0.0
        S = 575 + args.text_len
    else:
        fwd_s, bwd_s, sampled, S = cpu_layer_samples(args)
    if mode == "sampled":
        total = fwd_s + bwd_s
        return {"value": round(8.0 / total, 4), "unit": "frames/s", "cores": cores, "cpu_model": cpu, "kind": "port", "measured": False, "seconds_per_window": round(total, 1),
                "sample": (f"oracle fp32 at full dims, one 8-frame window: 1 CLIP layer fwd, 1 windowed + 1 global SAM block fwd+bwd, "
                           f"1 SAM adapter fwd+bwd, 1 LLaMA layer (S={S}) fwd+dgrad; {sampled:.1f} s measured, scaled by layer counts "
                           f"(stems/projector/decoder/lm_head <2% of FLOPs, not included) to {total:.0f} s per window")}
    from oracle.lazy_weights import LazyRoundedWeights
    bf = torch.bfloat16
    batch = synthetic_batch(d, B=1, T=8, L=args.text_len, n_det=3, seed=7)
    kw = batch.as_kwargs(inference=(mode != "train"))
    for k in ("global_enc_images", "grounding_enc_images"):
        kw[k] = kw[k].to(bf).float()
    if mode == "train":
        from grove_amd.model.GROVE import trainable_names
        from grove_amd.synthetic import synthetic_state_dict
        names = set(trainable_names(d))
        sd = {k: v.to(bf).float().cpu().requires_grad_(k in names)
              for k, v in synthetic_state_dict(d, device=dev if dev is not None else "cpu", dtype=torch.float32).items()}
        t0 = time.perf_counter()
        out = O.model_forward(sd, d, **kw)
        t1 = time.perf_counter()
        out["loss"].backward()
        t2 = time.perf_counter()
        total = t2 - t0
        c1 = None
        try:  # SURVEY section 8(d)'s other CPU figure: config 1 (forward + 64-token greedy decode), beside the fwd + bwd window
            del out
            c1 = cpu_config1_leg(sd, d, t1 - t0, cores)
        except Exception as e:
            c1 = {"error": repr(e)}
        return {"value": round(8.0 / total, 4), "unit": "frames/s", "cores": cores, "cpu_model": cpu, "kind": "port", "measured": True, "seconds_per_window": round(total, 1),
                "forward_seconds_measured": round(t1 - t0, 1), "backward_seconds_measured": round(t2 - t1, 1), "config1_infer_greedy64": c1,
                "sample": (f"oracle fp32 at full dims, ONE whole 8-frame window of the step (B=1, T=8, L={args.text_len}), forward {t1 - t0:.1f} s + "
                           f"backward through torch autograd {t2 - t1:.1f} s, timed end to end on {cores} threads")}
    sd = LazyRoundedWeights(d, gen_device=dev if dev is not None else "cpu")
    t0 = time.perf_counter()
    with torch.no_grad():
        O.model_forward(sd, d, **kw)
    t_fwd = time.perf_counter() - t0 - sd.fetch_seconds
    total = t_fwd + bwd_s
    return {"value": round(8.0 / total, 4), "unit": "frames/s", "cores": cores, "cpu_model": cpu, "kind": "port", "measured": True, "seconds_per_window": round(total, 1),
            "forward_seconds_measured": round(t_fwd, 1), "forward_frames_per_s": round(8.0 / t_fwd, 4), "backward_seconds_from_layer_samples": round(bwd_s, 1),
            "sample": (f"oracle fp32 at full dims, ONE whole 8-frame window of the step (B=1, T=8, L={args.text_len}): the full forward (SAM + CLIP + projector "
                       f"+ LLaMA + box decoder) run and timed end to end = {t_fwd:.1f} s on {cores} threads (weight generation, {sd.fetch_seconds:.1f} s, excluded; "
                       f"the per-layer samples put it at {fwd_s:.1f} s); the step's backward added from per-layer fwd+bwd samples x layer counts = {bwd_s:.1f} s "
                       f"(`--cpu_baseline train` times the whole fwd+bwd instead)")}


def tiny_box_l1(dev):
    """Box L1 (normalised cxcywh) of the HIP path vs the CPU oracle on the tiny-dims golden case."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    bf = torch.bfloat16
    sd = synthetic_state_dict(TINY)
    model = GROVEForCausalLM(dims=TINY, device=dev, state_dict=sd, det_token_idx=TINY.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    batch = synthetic_batch(TINY, B=2, T=8, L=40, n_det=3, seed=2)
    kw = batch.as_kwargs(inference=True)
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
        kw[k] = kw[k].to(bf).float()
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    out = model(**kd)
    with torch.no_grad():
        ref = O.model_forward({k: v.to(bf).float() for k, v in sd.items()}, TINY, **kw)
    return float((out["flat_boxes"].cpu() - ref["flat_boxes"]).abs().mean())


def tiny_train_parity(dev):
    """SURVEY.md section 8(d)'s accuracy figures for the training configuration, on the tiny-dims case the parity tests use:
    objectness-logit error, relative error of the loss terms and the cosine of the whole trainable gradient against torch
    autograd through the fp32 CPU oracle (same bf16-rounded weights and inputs)."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    bf = torch.bfloat16
    d = TINY
    sd = synthetic_state_dict(d)
    names = trainable_names(d)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, train=True)
    batch = synthetic_batch(d, B=2, T=8, L=48, n_det=2, seed=1, ragged=True)
    kw = batch.as_kwargs()
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
        kw[k] = kw[k].to(bf).float()
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    model.zero_grad()
    out = model(**kd)
    model.backward(out["loss"])
    sdg = {k: v.to(bf).float().requires_grad_(k in names) for k, v in sd.items()}
    ref = O.model_forward(sdg, d, **kw)
    ref["loss"].backward()
    gs, rs = [], []
    for n in names:
        g, r = model._grad[n].detach().float().cpu(), sdg[n].grad
        if n.endswith("conv3d.weight"):
            g = g.view(r.shape[0], 3, 3, 3, r.shape[1]).permute(0, 4, 1, 2, 3)
        gs.append(g.reshape(-1))
        rs.append(r.reshape(-1))
    g, r = torch.cat(gs), torch.cat(rs)
    terms = ("ce_loss", "giou_loss", "l1_loss", "temp_objectness_loss")
    return {"box_l1_train_mode": round(float((out["flat_boxes"].detach().cpu() - ref["flat_boxes"].detach()).abs().mean()), 6),
            "loss_rel_err_max": round(max(abs(float(out[k]) - float(ref[k])) / max(abs(float(ref[k])), 1e-12) for k in terms), 5),
            "objectness_logit_abs_err": round(float((out["flat_logits"].detach().cpu() - ref["flat_logits"].detach()).abs().max()), 5),
            "grad_cosine": round(float(torch.nn.functional.cosine_similarity(g, r, dim=0)), 5),
            "grad_norm_ratio": round(float(g.norm() / r.norm()), 4), "trainable_elements": int(g.numel())}


class Progress:
    """Per-rank stage marker + stall watchdog. `stage(name, limit_s)` prints the stage to stderr, writes it to
    $GROVE_BENCH_PROGRESS_DIR/rank<r> (read by the self-launching parent when it has to kill a stalled run) and re-arms
    faulthandler: a rank that sits in one stage longer than `limit_s` dumps the Python stack of every thread — i.e. WHICH call it
    sat in — to stderr and exits non-zero, so a wedged first 8-GPU run ends in minutes with a record instead of burning the
    driver's whole time limit. (RCCL's own watchdog reports a stuck collective after the process-group timeout, 120 s.)"""

    def __init__(self, rank):
        self.rank, self.dir = rank, os.environ.get("GROVE_BENCH_PROGRESS_DIR")
        self.t0 = time.time()

    def stage(self, name, limit_s=300):
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
        msg = f"[bench rank {self.rank}] +{time.time() - self.t0:6.1f}s stage: {name}"
        print(msg, file=sys.stderr, flush=True)
        if self.dir:
            try:
                with open(os.path.join(self.dir, f"rank{self.rank}"), "w") as fh:
                    fh.write(f"{time.time():.1f} {name}\n")
            except OSError:
                pass
        if limit_s:
            faulthandler.dump_traceback_later(limit_s, exit=True, file=sys.stderr)

    def done(self):
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
        self.stage("done", limit_s=0)


def self_launch(n, argv, timeout_s=1500):
    """`python bench.py --gpus N` without a launcher around it: start N ranks (one per GPU) with torch.distributed.run as a CHILD
    process (its own session / process group) and pass its single JSON line through. This parent never touches the GPU (no HIP
    call, no torch.cuda query), so nothing is re-exec'ed from a GPU-initialised process; the child's exit code becomes ours. If the
    child has not finished after `timeout_s` the parent prints every rank's last stage (Progress files), kills the child's process
    group — exactly the processes this call started — and exits 124."""
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pdir = tempfile.mkdtemp(prefix="grove_bench_progress_")
    env["GROVE_BENCH_PROGRESS_DIR"] = pdir
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        now = time.time()
        print(f"[bench] {n}-rank run exceeded {timeout_s} s; last stage of every rank:", file=sys.stderr)
        for r in range(n):
            try:
                ts, name = open(os.path.join(pdir, f"rank{r}")).read().strip().split(" ", 1)
                print(f"[bench]   rank {r}: '{name}' for {now - float(ts):.0f} s", file=sys.stderr)
            except (OSError, ValueError):
                print(f"[bench]   rank {r}: never reported a stage (stuck before / in the rendezvous)", file=sys.stderr)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        rc = 124
    finally:
        import shutil
        shutil.rmtree(pdir, ignore_errors=True)
    return rc


def attention_flops(dims, windows, frames, S, train=True):
    """Executed attention FLOPs of one step (2 x MAC; 4 L_q L_k d forward, 10 L_q L_k d backward per (batch, head); causal halves
    LLaMA's): CLIP forward (23 layers), LLaMA fwd + bwd, SAM windowed blocks on the REAL tokens (1024 queries x 196 keys per frame
    and head) and global blocks, backward for blocks after the first adapter (blocks 8-31)."""
    d = dims
    bwd = 2.5 if train else 0.0
    tok = d.clip_tokens
    clip = frames * (d.clip_layers - 1) * d.clip_heads * 4.0 * tok * tok * (d.clip_dim // d.clip_heads)
    llama = windows * d.n_layers * d.n_heads * 2.0 * S * S * d.head_dim * (1.0 + bwd)
    g2 = d.sam_grid ** 2
    hd = d.sam_dim // d.sam_heads
    first_grad = min(d.sam_global) + 1
    n_win = d.sam_depth - len(d.sam_global)
    n_win_bwd = sum(1 for i in range(first_grad, d.sam_depth) if i not in d.sam_global)
    n_glob_bwd = sum(1 for i in d.sam_global if i >= first_grad)
    win = frames * d.sam_heads * 4.0 * g2 * d.sam_window ** 2 * hd * (n_win + bwd * n_win_bwd)
    glob = frames * d.sam_heads * 4.0 * g2 * g2 * hd * (len(d.sam_global) + bwd * n_glob_bwd)
    return clip + llama + win + glob


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def host_memory_available_gb():
    """min(MemAvailable, the cgroup's remaining allowance) in GB — what an in-process CPU baseline may use without being killed."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) / 1e6
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            m = open(lim).read().strip()
            if m != "max":
                left = (int(m) - int(open(cur).read().strip())) / 1e9
                avail = left if avail is None else min(avail, left)
        except (OSError, ValueError):
            pass
    return avail if avail is not None else 0.0


def infer_box_l1(dev, dtype):
    """Box L1 of the inference configuration of this mode (bf16 or fp8 linear layers) against the fp32 CPU oracle on the tiny-dims
    case of tests/test_parity_r2_gpu.py::test_fp8_vit_llama_path_vs_oracle_and_bf16 (every GEMM K a multiple of 128)."""
    import dataclasses
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    bf = torch.bfloat16
    d = dataclasses.replace(TINY, clip_dim=128, clip_heads=2, clip_mlp=256)
    sd = synthetic_state_dict(d)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, gemm_dtype=dtype)
    batch = synthetic_batch(d, B=2, T=8, L=40, n_det=3, seed=2)
    kw = batch.as_kwargs(inference=True)
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
        kw[k] = kw[k].to(bf).float()
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    out = model(**kd)
    with torch.no_grad():
        ref = O.model_forward({k: v.to(bf).float() for k, v in sd.items()}, d, **kw)
    return float((out["flat_boxes"].cpu() - ref["flat_boxes"]).abs().mean())


def infer_bench(args, dev, dims, world=1, rank=0):
    """BASELINE config 5 as an explicit mode (`--mode infer`; never the default line): clip inference with the SAM mask decoder at
    T = 32 — CLIP + LLaMA linear layers on the fp8 MFMA GEMM with `--dtype fp8` (bf16 otherwise), SAM tower, box decoder AND the mask
    branch (low-res masks -> postprocess to the original frame size) for every ([DET], frame) instance. One step = one batch of
    `--batch` clips x `--frames` frames (= frames / 8 independent windows each) through model_forward(inference=True) + predict_masks."""
    from grove_amd import GROVEForCausalLM, ops
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    bf = torch.bfloat16
    sd = synthetic_state_dict(dims, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=dims, device=dev, state_dict=sd, det_token_idx=dims.det_token_idx, num_frames=8, gemm_dtype=args.dtype,
                             fp8_policy=args.fp8_policy)
    del sd
    torch.cuda.empty_cache()
    # N > 1 (config 5 is an 8-GPU job): replicas only — every rank runs its own `--batch` clips (seeded by rank, the
    # DistributedSampler shard of infer_iground.py:538-551), no data-path collective; one all_gather_object of the per-rank
    # results at the end of the run, as infer_iground.py:290-293
    b = synthetic_batch(dims, B=args.batch, T=args.frames, L=args.text_len, n_det=3, seed=7 + 1000 * rank, device=dev, dtype=bf)
    kw = b.as_kwargs(inference=True)
    band = int(dims.sam_image * 360 / 640)
    rec = []
    orig_fp8, orig_lin = ops.linear_fp8, ops.gemm_raw

    def step(instrument=False):
        out = model(**kw)
        n_inst = out["flat_boxes"].shape[0]
        frames = args.batch * args.frames
        per = n_inst // frames
        inst_frame = torch.arange(frames, dtype=torch.int32, device=dev).repeat_interleave(per)
        text = model._last_text if hasattr(model, "_last_text") else None
        res = model.predict_masks(out["image_embeddings"], text, inst_frame, input_size=(band, dims.sam_image), original_size=(360, 640))
        return out, res

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gathered = 1
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        # the job's one collective (after the timed region, like the reference's end-of-run gather): per-clip results to every rank
        from grove_amd.infer import update_and_sort_video_outputs
        mine = {f"rank{rank}_clip{c}": {"pred_bboxes": [x.cpu() for x in out["pred_bboxes"][c]],
                                         "logits_temp_objectness": [x.cpu() for x in out["logits_temp_objectness"][c]]} for c in range(args.batch)}
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        gathered = len(update_and_sort_video_outputs(parts)) // args.batch
    # dominant kernel of the mode: event-timed launches of the ViT / LLaMA linear layers in one extra pass
    def timed_fp8(x, wq, ws, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_fp8(x, wq, ws, *a, **k)
        e1.record()
        M = (k["xq"][0] if k.get("xq") is not None else x).shape[0]
        rec.append((e0, e1, 2.0 * M * wq.shape[0] * wq.shape[1]))
        return r
    if args.dtype == "fp8":
        ops.linear_fp8 = timed_fp8
        overlap, model.tower_overlap = model.tower_overlap, False  # a kernel is priced on its own (no SAM tower beside it on a second stream)
        try:
            step()
            torch.cuda.synchronize()
        finally:
            ops.linear_fp8 = orig_fp8
            model.tower_overlap = overlap
    frames = world * args.batch * args.frames * args.steps
    line = {"metric": "frames/sec (clip inference fwd + SAM masks)", "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"GROVE inference + SAM mask decoder: {args.batch} clips x T={args.frames} frames ({args.batch * args.frames // 8} windows), "
                                   f"LLaVA-1.5-7B + CLIP ViT-L/14-336 ({args.dtype} linear layers) + SAM ViT-H@512 (bf16) + box / mask decoder, text L={args.text_len}",
                       "dims": args.dims, "fp8_policy": args.fp8_policy if args.dtype == "fp8" else None, "instances": int(out["flat_boxes"].shape[0]), "mask_shape": list(res["masks"].shape),
                       "parallelism": f"dp{world} (replicas only: clips sharded over ranks, end-of-run all_gather_object)", "ranks": world,
                       "ranks_gathered": gathered, "collective_backend": (dist.get_backend() if world > 1 else None),
                       "frames_per_sec_per_gpu": round(frames / dt / world, 3)}}
    if rec:
        fl = sum(f for _, _, f in rec)
        secs = sum(e0.elapsed_time(e1) for e0, e1, _ in rec) * 1e-3
        line["roofline"] = {"bound": "mfma", "achieved": round(fl / secs / 1e12, 2), "peak": 5000.0, "unit": "TFLOP/s", "frac": round(fl / secs / 1e12 / 5000.0, 4),
                            "traffic": None, "kernel": "gemm_nt_pp_kernel<.., FP8> (grove_gemm_fp8: the e4m3 instances of the pipelined kernel; launch + its activation quantisation)", "launches_per_step": len(rec),
                            "flops_per_step": fl}
    return line


def infer_iground_bench(args, dev, dims, world=1, rank=0):
    """BASELINE config 2 as the reference RUNS it (`--mode infer_iground`; VERDICT r4 missing #2): infer_iground.py:150-288 per clip —
    48 frames -> six 8-frame windows; the centre window through `evaluate` with `max_tokens_new=64` greedy tokens (prefill of
    575 + prompt positions, then 64 cached decode steps), the other five windows teacher-forced with the generated answer — once
    with the reference's batch 1 (`infer_clip`, infer_iground.py:49-51) and once clip-batched (`infer_clips_batched`, up to 8 clips
    share one encode, one prefill and ONE weight stream per generated token). One step = `--clips` clips. eos is disabled so that
    every clip decodes all 64 tokens (the longest, fully deterministic case; synthetic weights would stop at arbitrary places)."""
    from dataclasses import replace
    from grove_amd import GROVEForCausalLM
    from grove_amd.infer import infer_clip, infer_clips_batched
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    bf = torch.bfloat16
    sd = synthetic_state_dict(dims, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=dims, device=dev, state_dict=sd, det_token_idx=dims.det_token_idx, num_frames=8)
    del sd
    torch.cuda.empty_cache()
    model.dims = replace(model.dims, eos_token_id=-1)
    F, N, new = args.frames, args.clips, args.max_new_tokens
    clips = []
    for c in range(N):
        b = synthetic_batch(dims, B=1, T=F, L=24, n_det=2, seed=11 + c + 1000 * rank, device=dev, dtype=bf)
        clips.append((b.global_enc_images, b.grounding_enc_images, b.original_size_list[0]))
    prompt = synthetic_batch(dims, B=1, T=F, L=24, n_det=2, seed=11, device="cpu").input_ids[0, :20].clone()

    def run_b1():
        return [infer_clip(model, g, s_, prompt, sz, max_tokens_new=new) for g, s_, sz in clips]

    def run_batched(times=None):
        out = []
        for i in range(0, N, args.clips_per_batch):
            out += infer_clips_batched(model, clips[i:i + args.clips_per_batch], prompt, max_tokens_new=new, stage_times=times)
        return out

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        for _ in range(args.steps):
            res = fn()
            marks.append(time.perf_counter() - t0)  # (host time: the results of a step are read back inside it, so this is the step's end)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        timed.steps_s = [round(b_ - a, 4) for a, b_ in zip([0.0] + marks[:-1], marks)]
        return dt, res
    dt_b, res_b = timed(run_batched)
    steps_batched_s = timed.steps_s
    dt_1, res_1 = timed(run_b1)
    same_ids = all(torch.equal(a["output_ids"], b_["output_ids"]) for a, b_ in zip(res_b, res_1))
    box_diff = 0.0
    for a, b_, (_, _, sz) in zip(res_b, res_1, clips):
        for x, y in zip(a["pred_bboxes"], b_["pred_bboxes"]):
            if x.shape == y.shape and x.numel():
                box_diff = max(box_diff, float((x.float().cpu() - y.float().cpu()).abs().max()) / max(sz))
    # by construction (round 6): the batched run is batch-invariant (infer_clips_batched's default, model.batch_invariant_mode) — every clip
    # alone, as a batch of ONE under the same mode, must return the same ids and the same boxes bit for bit
    ones = [infer_clips_batched(model, [c], prompt, max_tokens_new=new)[0] for c in clips[:min(N, 3)]]
    invariant = all(torch.equal(a["output_ids"], b_["output_ids"]) and
                    all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a["pred_bboxes"], b_["pred_bboxes"])) for a, b_ in zip(res_b, ones))
    stages = {}
    run_batched(stages)  # one extra pass with a device sync around every stage: the breakdown (not part of the timed region)
    # decode alone at the batch the job decodes at: per-token time and the weight stream's share of the HBM peak
    with torch.no_grad():
        nb = min(args.clips_per_batch, N)
        g_c = torch.cat([g[:, :, :8] for g, _, _ in clips[:nb]], 0).contiguous()
        feats, _ = model.encode_images(g_c)
        pr = prompt[None].repeat(nb, 1).to(dev)

        def gen(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(input_ids=pr, image_features=feats, max_new_tokens=n, eos_token_id=-1)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        gen(3)
        t9, t57 = min(gen(9), gen(9)), min(gen(57), gen(57))
        per_tok = (t57 - t9) / 48
        prefill_s = max(t9 - 8 * per_tok, 0.0)
    wbytes = 2.0 * (dims.n_layers * (4 * dims.hidden * dims.hidden + 3 * dims.hidden * dims.mlp) + dims.vocab * dims.hidden)
    frames = world * N * F * args.steps
    per_step = dt_b / args.steps
    line = {"metric": "frames/sec (infer_iground: sliding-window clip inference, 64 greedy tokens per clip)", "value": round(frames / dt_b, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_step * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"infer_iground.py clip inference: {N} clips/GPU x {F} frames ({F // 8} windows each: centre window evaluate + {new} greedy tokens, "
                                   f"{F // 8 - 1} windows teacher-forced), LLaVA-1.5-7B + CLIP ViT-L/14-336 + SAM ViT-H@512 + box decoder, prompt of 20 ids",
                       "dims": args.dims, "clips_per_batch": args.clips_per_batch, "max_new_tokens": new, "eos": "disabled (every clip decodes all tokens)",
                       "parallelism": f"dp{world} (replicas only: clips sharded over ranks)", "ranks": world,
                       "clips_per_s": round(world * N * args.steps / dt_b, 3), "clips_per_s_batch1_reference_form": round(world * N * args.steps / dt_1, 3),
                       "speedup_over_batch1": round(dt_1 / dt_b, 3), "ids_equal_to_batch1": bool(same_ids), "max_box_diff_vs_batch1_normalised": box_diff,
                       "batch_invariant": True, "bit_identical_to_one_clip_batches": bool(invariant), "seconds_of_each_timed_step": steps_batched_s,
                       "stage_seconds_per_step_batched": {k: round(v, 4) for k, v in stages.items()},
                       "stage_note": "encode = CLIP + SAM towers of the centre windows; evaluate = prefill + greedy decode + box decoder; windows = the other windows' "
                                     "forward (all clips of a batch in one launch sequence); measured in one extra pass with device syncs around the stages"},
            "roofline": {"bound": "hbm", "kernel": (f"gemv_mfma_kernel<., 8>" if nb >= 3 else f"gemv_kernel<{nb}, .>") + f" (grove_gemv_bf16: the cached decode step's weight stream, {nb} sequences per launch)",
                         "achieved": round(wbytes / per_tok / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(wbytes / per_tok / 8e12, 4), "traffic": None,
                         "ms_per_token": round(per_tok * 1e3, 3), "sequences_per_token_step": nb, "weight_bytes_per_token_step": wbytes,
                         "prefill_ms": round(prefill_s * 1e3, 2), "decode_share_of_step": round(min(1.0, (N / nb) * new * per_tok / per_step), 3)}}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--text_len", type=int, default=128)
    ap.add_argument("--dims", default="full", choices=["full", "tiny"])
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_baseline", default="auto", choices=["auto", "window", "train", "sampled"],
                    help="CPU oracle leg: `train` = one whole window fwd+bwd through torch autograd, timed end to end (needs ~80 GB of host "
                         "memory; what `auto` picks when the host has >= 128 GB available), `window` = the whole window forward timed + the "
                         "backward from per-layer samples, `sampled` = per-layer samples only")
    ap.add_argument("--launch_timeout", type=int, default=1500, help="--gpus N self-launch: seconds before the parent kills a stalled run")
    ap.add_argument("--stage_timeout", type=int, default=300, help="seconds a rank may sit in one stage before it dumps its stacks and exits")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "infer_iground"],
                    help="train (default, the headline line: BASELINE config 3), infer (config 5: inference + SAM masks, use --frames 32 --dtype fp8) or "
                         "infer_iground (config 2 as infer_iground.py runs it: 48-frame clips, centre-window evaluate with 64 greedy tokens, clip-batched)")
    ap.add_argument("--clips", type=int, default=8, help="--mode infer_iground: clips per step")
    ap.add_argument("--clips_per_batch", type=int, default=8, help="--mode infer_iground: clips decoded together (1 = the reference's batch-1 form)")
    ap.add_argument("--max_new_tokens", type=int, default=64, help="--mode infer_iground: greedy tokens per clip (infer_iground.py:192)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"], help="--mode infer: linear layers of the CLIP tower and the LLaMA stack")
    ap.add_argument("--fp8_policy", default="sam_mlp", choices=["sam_mlp", "all", "det16_kv16", "det16_kv16_clip16"],
                    help="--mode infer --dtype fp8: which GEMMs / rows stay bf16 (DESIGN section 8; _clip16 = the CLIP tower in bf16)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "allreduce", "rs_ag", "a2a_f32"],
                    help="N > 1: one all-reduce per gradient bucket, reduce-scatter + all-gather per bucket, or all-to-all + fp32 sum + all-gather; "
                         "auto (default) = all-reduce for the line's timed region; the calibration pass that FOLLOWS it times every arm, prints the table, and re-times "
                         "the line on an arm that beats all-reduce by more than 1 %% (a stalled calibration cannot lose the line)")
    ap.add_argument("--no_calibration", action="store_true", help="N > 1: skip the exchange-arm calibration pass (then --exchange auto = allreduce)")
    ap.add_argument("--calibration_steps", type=int, default=3)
    ap.add_argument("--via_train_loop", action="store_true",
                    help="time grove_amd.train.train() (the reference's train.py entry point: loss meters, logging cadence) instead of the bare "
                         "engine loop, and report both (VERDICT r4 next #9b: the entry point must be within 1 %% of the engine loop)")
    ap.add_argument("--dense_embed", action="store_true",
                    help="N > 1: exchange embed_tokens' gradient densely (131 M elements) instead of as touched rows (A/B arm)")
    ap.add_argument("--no_comm_overlap", action="store_true",
                    help="N > 1: exchange all gradients after the backward instead of group by group from inside it (exposed-communication A/B)")
    ap.add_argument("--gemm_blocks", type=int, default=0,
                    help="N > 1 A/B: resident blocks of the persistent GEMMs (0 = one per CU); fewer leaves CUs to the overlapped RCCL kernels")
    ap.add_argument("--clip_alpha", type=float, default=0.0,
                    help="--mode train: alpha of the CLIP tower's 8 Conv3d adapters (default 0 = SURVEY section 8(d)'s synthetic weights, whose adapters are "
                         "skipped; a trained checkpoint has them active — NOT the headline configuration, an A/B aid)")
    ap.add_argument("--stream", default="default", choices=["default", "fp32", "bf16"],
                    help="--mode train: residual streams of the three towers (default = bf16, what the reference stores; fp32 = the inference models' form)")
    ap.add_argument("--serial_towers", action="store_true",
                    help="run the SAM tower on the main stream as well (no kernel overlap): how profiles/*_kernel_stats are collected")
    args = ap.parse_args()
    if args.gemm_blocks:
        os.environ["GROVE_GEMM_BLOCKS"] = str(args.gemm_blocks)  # (read when the library is loaded, in every rank)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))

    # stdout carries exactly ONE line, the JSON: everything else that writes to fd 1 (RCCL prints its version banner there, from
    # every rank, when the box exports NCCL_DEBUG=VERSION — and flushes it at exit, i.e. AFTER a normal print) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box: GROVE_BENCH_BACKEND=gloo GROVE_BENCH_ONE_GPU=1 runs every rank on cuda:0 and
    # moves the collectives through the host (RCCL refuses two ranks on one device); never set by the driver
    if os.environ.get("GROVE_BENCH_ONE_GPU"):
        local = 0
    prog = Progress(rank)
    prog.stage("rendezvous + process group", args.stage_timeout)
    rccl_ranks = None
    if world > 1:
        from grove_amd.train import init_distributed
        # 120 s: a collective that has not completed by then is wedged (a bucket is 128 MB); RCCL's watchdog then reports WHICH one
        init_distributed(local, timeout_s=120, backend=os.environ.get("GROVE_BENCH_BACKEND", "nccl"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        prog.stage("first collective (all-reduce of ones)", args.stage_timeout)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())  # how many ranks the collective backend really summed over
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)
    from grove_amd.synthetic import FULL, TINY
    dims = FULL if args.dims == "full" else TINY

    if args.mode == "infer_iground":
        if args.frames == 16:
            args.frames = 48  # (the reference's clip length; --frames overrides)
        prog.stage("infer_iground bench", 4 * args.stage_timeout)
        line = infer_iground_bench(args, dev, dims, world, rank)
        line["config"]["rccl_ranks"] = rccl_ranks
        prog.done()
        if rank == 0:
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.mode == "infer":
        prog.stage("inference bench", 3 * args.stage_timeout)
        line = infer_bench(args, dev, dims, world, rank)
        line["config"]["rccl_ranks"] = rccl_ranks
        prog.done()
        if rank == 0:
            if not args.no_cpu_baseline and world == 1:
                try:  # parity figure of THIS mode's arithmetic (fp8: the quantised path's own figure)
                    line["box_l1_vs_oracle"] = {"value": round(infer_box_l1(dev, args.dtype), 6), "case": "tiny dims (K % 128 == 0), B=2, T=8, vs fp32 CPU oracle",
                                                "tolerance": 1e-3 if args.dtype == "bf16" else FP8_BOX_L1_BOUND}
                except Exception as e:
                    line["box_l1_vs_oracle"] = {"error": repr(e)}
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    prog.stage("build model + engine (weights, broadcast)", 2 * args.stage_timeout)
    model, engine = build(dims, dev, args)
    model.tower_overlap = not args.serial_towers
    batch = make_batch(dims, dev, args, rank)

    def step():
        out = engine(**batch)
        engine.backward(out["loss"])
        engine.step()
        return out

    prog.stage(f"warm-up ({args.warmup} steps)", args.stage_timeout)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    calibration = None  # (N > 1: filled AFTER the line is complete, see the epilogue at the end of main)
    args.exchange_asked = args.exchange
    if engine.exchange is not None:
        args.exchange = engine.exchange.mode
    prog.stage("barrier before the timed region", args.stage_timeout)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prog.stage(f"timed region ({args.steps} steps)", args.stage_timeout + 2 * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prog.stage("max over ranks", args.stage_timeout)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    train_entry = None
    if args.via_train_loop:
        # the same K steps through grove_amd.train.train() — the reference's train.py hot loop (meters of the loss terms, logging
        # cadence, batch_time) — bracketed the same way: north_star names THIS entry point as the API surface
        import copy
        from grove_amd.train import train as train_entry_point
        prog.stage(f"train() entry point ({args.warmup} + {args.steps} steps)", args.stage_timeout + 2 * args.steps)
        targs = copy.copy(engine.args)
        targs.grad_accumulation_steps, targs.print_freq = 1, max(args.steps, 1)

        def endless():
            while True:
                yield batch
        it = endless()
        targs.steps_per_epoch = max(args.warmup, 1)
        it = train_entry_point(it, engine, 0, targs, log=lambda m: None)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        targs.steps_per_epoch = args.steps
        logged = []
        t1 = time.perf_counter()
        train_entry_point(it, engine, 1, targs, log=logged.append)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t[0])
        train_entry = {"ms_per_step": round(dt2 / args.steps * 1e3, 2), "engine_loop_ms_per_step": round(dt / args.steps * 1e3, 2),
                       "ratio": round(dt2 / dt, 4), "logged_line": (logged[0] if logged else None),
                       "what": "grove_amd.train.train(data_iter, engine, epoch, args): the reference's hot loop (train.py:704-793) incl. its loss "
                               "meters; loss terms are read back one micro-step late through a pinned buffer, never between forward and backward"}
    frames = world * args.batch * args.frames * args.steps
    loss = float(out["loss"])
    exposed_ms = engine.exposed_comm_ms() if world > 1 else None  # last timed step: what the compute stream waited for the collectives
    capped_launches = engine.exchange.reserved_launch_polls if (world > 1 and engine.exchange is not None) else None

    prog.stage("instrumented step (per-kernel HIP events)", args.stage_timeout)
    per_kernel = instrumented_gemm_pass(engine, batch)
    dom = max(per_kernel, key=lambda k: per_kernel[k][2])  # the kernel that takes most of the step
    n_launch, flops, secs = per_kernel[dom]
    all_n, all_f, all_s = (sum(v[i] for v in per_kernel.values()) for i in range(3))
    traffic = None
    try:  # memory-side bytes per launch from the committed PMC passes (profiles/, collected as the microarch guide prescribes)
        rec = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_gemm_traffic.json"))[-1]  # the latest round's
        with open(os.path.join(ROOT, "profiles", rec)) as fh:
            traffic = json.load(fh)["launch_weighted_mean_bytes"].get(dom[dom.index("<"):dom.index(",")] + ">")
    except Exception:
        pass
    if rank == 0:
        res = {
            "metric": "frames/sec (T=16 clip fwd+bwd)", "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"train.py iGround fine-tune step: {args.batch} clips/GPU x T={args.frames} frames "
                                   f"({args.batch * args.frames // 8} independent 8-frame windows), LLaVA-1.5-7B + CLIP ViT-L/14-336 + "
                                   f"SAM ViT-H@512 + box decoder, text L={args.text_len}, fwd+bwd+AdamW, shipped freeze policy",
                       "dims": args.dims, "global_batch_clips": world * args.batch, "frames_per_clip": args.frames,
                       "parallelism": f"dp{world}", "ranks": world,
                       "gradient_exchange": (None if world == 1 else f"{args.exchange}, bf16 wire, " +
                                             ("after the backward" if args.no_comm_overlap else "overlapped with the backward (per parameter group)") +
                                             (", embed_tokens dense" if args.dense_embed else ", embed_tokens as touched rows (all-gather of ids + rows, fp32 sum)")),
                       "exposed_comm_ms": (None if exposed_ms is None else round(exposed_ms, 3)),
                       "exchange_calibration": calibration, "train_entry_point": train_entry,
                       "gemm_persistent_blocks": args.gemm_blocks or ("one per CU" if world == 1 else
                                                                      f"one per CU; CUs - {engine.exchange.reserve_cus} while gradient buckets are in flight "
                                                                      f"({capped_launches} GEMM launches under the cap over the run)"),
                       "collective_backend": (dist.get_backend() if world > 1 else None), "rccl_ranks": rccl_ranks,
                       "rccl_max_channels": (os.environ.get("NCCL_MAX_NCHANNELS") if world > 1 else None),
                       "frames_per_sec_per_gpu": round(frames / dt / world, 3), "last_loss": round(loss, 4),
                       "towers": "serial" if args.serial_towers else "SAM tower on a second stream beside CLIP->LLaMA (roofline: one extra step with the towers serialised)"},
            "roofline": {"bound": "mfma", "achieved": round(flops / secs / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / secs / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "kernel": dom + " (grove_gemm_bf16)", "launches_per_step": n_launch,
                         "avg_launch_us": round(secs / n_launch * 1e6, 2), "flops_per_launch": round(flops / n_launch),
                         "share_of_step": round(secs / (dt / args.steps), 3),
                         "all_gemm_kernels": {"launches_per_step": all_n, "flops_per_step": all_f, "achieved": round(all_f / all_s / 1e12, 2),
                                              "share_of_step": round(all_s / (dt / args.steps), 3)}},
        }
        S_seq = 575 + args.text_len
        attn_f = attention_flops(dims, args.batch * args.frames // 8, args.batch * args.frames, S_seq)
        step_s = dt / args.steps
        alg_f = all_f + instrumented_gemm_pass.algorithmic_extra
        res["step_roofline"] = {"bound": "mfma", "achieved": round((alg_f + attn_f) / step_s / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": round((alg_f + attn_f) / step_s / 1e12 / PEAK_BF16_TFLOPS, 4),
                                "executed": {"achieved": round((all_f + attn_f) / step_s / 1e12, 2), "frac": round((all_f + attn_f) / step_s / 1e12 / PEAK_BF16_TFLOPS, 4)},
                                "flops_per_step": {"gemm_algorithmic": alg_f, "gemm_executed": all_f, "attention_executed": attn_f},
                                "note": "ALGORITHMIC GEMM FLOPs of the instrumented step (grove_gemm_bf16 launches; the Conv3d adapters at their direct-convolution "
                                        "count, 2 * rows * 27 C * C — they RUN in Winograd F(2x2x2, 3x3x3) form at 64 / 216 of it: `executed`) + attention FLOPs "
                                        "(4 LqLk d fwd, 10 bwd; real SAM tokens) / ms_per_step / peak: the whole step incl. norms, element-wise passes, optimizer — "
                                        "not only the dominant kernel. Weight-gradient (TN) GEMMs are not counted on either line"}
        if args.dims == "full":
            prog.stage("stage figures (ViT+LLaMA forward, decode)", 2 * args.stage_timeout)
            try:
                res["fused_vit_llama_forward"] = vit_llama_forward(model, batch, dims)
            except Exception as e:
                res["fused_vit_llama_forward"] = {"error": repr(e)}
            if world == 1:
                try:
                    res["greedy_decode"] = decode_rate(model, dims, dev)
                except Exception as e:
                    res["greedy_decode"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N = 1 figure (other N: GPU numbers only)
            try:  # recorded by the last `pytest -m gpu` run of tests/test_full_depth_gpu.py (committed under profiles/): not measured in this run
                rec = {}
                have = sorted(os.listdir(os.path.join(ROOT, "profiles")))
                for key, suffix, field in (("box_l1_vs_oracle_full", "full_depth_parity_full.json", "box_l1_vs_oracle_full"),
                                           ("box_l1_full_width_3_seeds", "full_width_box_l1_seeds.json", "mean"),
                                           ("box_l1_deep_narrow_mean_4_seeds", "full_depth_box_l1_seeds.json", "mean"),
                                           ("train_mode_box_l1_full", "full_depth_training_parity_full.json", "box_l1_train_mode_vs_oracle"),
                                           ("train_mode_loss_rel_err_full", "full_depth_training_parity_full.json", "loss_terms_rel_err"),
                                           ("train_mode_loss_rel_err_deep_narrow", "full_depth_training_parity_deep_narrow.json", "loss_terms_rel_err"),
                                           ("train_mode_whole_gradient_deep_narrow", "full_depth_training_parity_deep_narrow.json", "whole_gradient"),
                                           ("greedy_ids_equal_full_size", "full_size_greedy_parity.json", "ids_equal"),
                                           ("box_l1_from_generated_rows_deep_narrow", "decode_rows_precision_deep_narrow.json", "box_l1_from_f32_decode_rows"),
                                           ("fp8_sam_mlp_box_l1_full_width_3_seeds", "full_width_box_l1_seeds.json", "fp8_sam_mlp")):
                    # the NEWEST round's record of each figure (profiles/rNN_<suffix>, VERDICT r5 next #7a); every figure carries the round and
                    # file it was recorded in (VERDICT r4 weak #1c): nothing here is measured by this run
                    for fn in reversed([f for f in have if f[0] == "r" and f[1:3].isdigit() and f[3] == "_" and f[4:] == suffix]):
                        with open(os.path.join(ROOT, "profiles", fn)) as fh:
                            doc = json.load(fh)
                        if field in doc:
                            v = doc[field]
                            rec[key] = {"value": v["mean"] if isinstance(v, dict) and "mean" in v else v, "recorded_in_round": fn.split("_")[0], "record": "profiles/" + fn}
                            break
                res["full_depth_parity_recorded"] = rec
            except Exception:
                pass
            prog.stage("parity figures + CPU baseline (oracle on the host cores)", 4 * args.stage_timeout)
            try:
                res["box_l1_vs_oracle_tiny"] = round(tiny_box_l1(dev), 6)  # inference model (fp32 streams, fp32 box path): the configuration boxes are emitted from
                res["train_parity_vs_oracle_tiny"] = tiny_train_parity(dev)  # the training-mode model the step above times (incl. its own box L1)
                res["cpu_baseline"] = cpu_baseline(args, dev)
            except Exception as e:  # the baseline is informational; never lose the measured line
                res["cpu_baseline"] = {"error": repr(e)}
    if rank != 0:
        res = None
    if engine.exchange is not None and not args.no_calibration:
        exchange_epilogue(args, engine, step, res, prog, world, rank, dev, real_stdout, frames_per_step=world * args.batch * args.frames)
    prog.done()
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def exchange_epilogue(args, engine, step, res, prog, world, rank, dev, real_stdout, frames_per_step):
    """N > 1 only, AFTER the line is complete (timed on the start arm: all-reduce without a CU reservation, or the arm asked for): the
    first multi-GPU run is one shot with default arguments, so it measures every exchange arm itself — and it must
    not be able to lose the number it already has. A heartbeat thread watches the calibration: if no step completes for 90 s (a wedged
    collective; RCCL's own watchdog would abort the process at 120 s) rank 0 prints the line as it stands, marked, and every rank leaves
    with exit code 0. If an arm beats the start arm by more than 1 % and no arm was asked for explicitly, the K steps are timed again on
    it and THAT becomes `value` (the start arm's figure stays in the table)."""
    import threading
    from grove_amd.train import EXCHANGE_ARMS, calibrate_exchange
    prog.stage(f"exchange calibration ({len(EXCHANGE_ARMS)} arms x {args.calibration_steps + 1} steps), the line is already safe", limit_s=0)
    beat = [time.time(), "start"]
    done = threading.Event()
    stall_s = float(os.environ.get("GROVE_BENCH_STALL_S", "90"))               # (tests shorten it)
    test_stall_arm = int(os.environ.get("GROVE_BENCH_TEST_STALL_ARM", "-1"))   # (tests: arm number that never returns)

    def watch():
        while not done.wait(5.0):
            if time.time() - beat[0] > stall_s:
                if rank == 0 and res is not None:
                    res["config"]["exchange_calibration"] = {"status": f"STALLED in '{beat[1]}' (no step for {stall_s:.0f} s): this line is the start arm's, measured before the "
                                                                       f"calibration; arms finished: {beat[2:] }"}
                    os.write(real_stdout, (json.dumps(res) + "\n").encode())
                else:
                    time.sleep(3.0)  # let rank 0 print first
                print(f"[bench rank {rank}] exchange calibration stalled in '{beat[1]}': leaving with the line measured before it", file=sys.stderr, flush=True)
                os._exit(0)
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    ex = engine.exchange
    start_arm = (ex.mode, ex.reserve_cus)

    def beating_step():
        out = step()
        beat[0] = time.time()
        return out

    def on_arm(mode, reserve):
        beat[1] = f"{mode}, {reserve} CUs reserved"
        if len(beat) - 2 == test_stall_arm:
            time.sleep(1e6)
    try:
        table, best = calibrate_exchange(engine, beating_step, steps=args.calibration_steps, on_arm=on_arm, done_arms=beat)
    except Exception as e:  # an arm that raises (an RCCL error, an out-of-memory plan) must not take the measured line with it
        print(f"[bench rank {rank}] exchange calibration failed in '{beat[1]}': {e!r}", file=sys.stderr, flush=True)
        if rank == 0 and res is not None:
            res["config"]["exchange_calibration"] = {"status": f"FAILED in '{beat[1]}': {e!r}; this line is the start arm's, measured before the calibration; "
                                                               f"arms finished: {beat[2:]}"}
        ex.mode, ex.reserve_cus = start_arm
        # a rank that raised must not walk into main()'s final barrier — the ranks still inside the arm's collective leave through
        # the heartbeat's os._exit(0) after `stall_s`, and this one would then wait for them until --launch_timeout and turn the run's exit
        # code into 124 although the line was printed. So: the line goes out here, and this rank leaves now with the promised exit code 0
        # (an exit, never a re-exec; the process group is torn down by the exit).
        if rank == 0 and res is not None:
            os.write(real_stdout, (json.dumps(res) + "\n").encode())
        sys.stderr.flush()
        done.set()
        os._exit(0)
    mine = [a for a in table if (a["exchange"], a["reserved_cus"]) == start_arm]
    start_row = mine[0] if mine else None
    calibration = {"arms": table, "fastest": best, "steps_per_arm": args.calibration_steps, "line_timed_on": {"exchange": start_arm[0], "reserved_cus": start_arm[1]},
                   "note": "ms/step = wall clock between barriers, max over ranks, run AFTER the line's own timed region; NCCL_MAX_NCHANNELS is read at "
                           "communicator creation and cannot be an arm (left to RCCL unless set in the environment)"}
    retime = (args.exchange_asked == "auto" and best is not None and start_row is not None and (best["exchange"], best["reserved_cus"]) != start_arm
              and best["ms_per_step"] < 0.99 * start_row["ms_per_step"])
    if retime:
        beat[1] = f"re-timing on {best['exchange']}, {best['reserved_cus']} CUs reserved"
        ex.mode, ex.reserve_cus = best["exchange"], best["reserved_cus"]
        beating_step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            beating_step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt2 = float(t[0])
        exposed2 = engine.exposed_comm_ms()
        if rank == 0:
            calibration["start_arm_line"] = {"value": res["value"], "ms_per_step": res["ms_per_step"], "exposed_comm_ms": res["config"]["exposed_comm_ms"]}
            res["value"] = round(frames_per_step * args.steps / dt2, 3)
            res["ms_per_step"] = round(dt2 / args.steps * 1e3, 2)
            res["config"]["frames_per_sec_per_gpu"] = round(res["value"] / world, 3)
            res["config"]["exposed_comm_ms"] = None if exposed2 is None else round(exposed2, 3)
            res["config"]["gradient_exchange"] = res["config"]["gradient_exchange"].replace(start_arm[0], best["exchange"], 1) + \
                f" (re-timed on the calibration's fastest arm, {best['reserved_cus']} CUs reserved)"
            calibration["line_timed_on"] = {"exchange": best["exchange"], "reserved_cus": best["reserved_cus"]}
    else:
        ex.mode, ex.reserve_cus = start_arm
    if rank == 0:
        res["config"]["exchange_calibration"] = calibration
    done.set()


if __name__ == "__main__":
    main()
