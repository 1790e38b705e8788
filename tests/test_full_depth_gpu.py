"""Full-depth, full-width parity figure: the REAL model — LLaMA 32 x 4096, CLIP ViT-L 24 layers, SAM ViT-H 32
blocks, 2-layer box decoder — B=1 clip x T=8 frames, text L=128 with three [DET], inference forward through the HIP path against the
fp32 CPU oracle on the same bf16-rounded synthetic weights and inputs. Depth is where bf16 error accumulates; every other GPU test
is either full depth at tiny width or full width at 1-3 layers.

The oracle's weights are produced lazily, one tensor at a time, from the same name-keyed generator the device weights come from
(grove_amd/synthetic.py: bit-identical on CPU and GPU), so the host never holds the 30 GB fp32 state dict. Slow (the oracle runs
~55 TFLOP on the host cores): about 2-4 minutes. The measured figures are written to gpurun_out/full_depth_parity.json (copied to
profiles/ by hand) before the assertions, so a miss is still reported.
"""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from oracle.lazy_weights import LazyRoundedWeights  # noqa: E402  (moved: bench.py's measured CPU baseline uses it too)


def tag(outliers):
    return f"_outliers{int(outliers)}" if outliers else ""


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12)).item()


def cos_norm64(a, b, chunk=1 << 24):
    """(cosine, |a| / |b|) of two flat fp32 tensors with every sum accumulated in FP64, chunk by chunk (VERDICT r5 weak #1c: an fp32
    cosine over 481 M elements read 1.0014 — a ruler whose own error is 1.4e-3 cannot gate at 0.99x)."""
    a, b = a.reshape(-1), b.reshape(-1)
    ab = aa = bb = 0.0
    for i in range(0, a.numel(), chunk):
        x, y = a[i:i + chunk].double(), b[i:i + chunk].double()
        ab += float(x @ y)
        aa += float(x @ x)
        bb += float(y @ y)
    return (ab / max((aa * bb) ** 0.5, 1e-300), (aa / max(bb, 1e-300)) ** 0.5)


def deep_narrow_dims():
    """Full DEPTH (32 LLaMA layers, 24 CLIP layers, 32 SAM blocks with the real global-block positions and 14x14 windows, real
    336 / 512-pixel inputs and token counts) at a quarter of the width: the oracle runs in seconds, and the error that matters here
    — bf16 rounding accumulated along the residual streams — grows with depth, not width."""
    import dataclasses
    from grove_amd.synthetic import FULL
    return dataclasses.replace(FULL, hidden=1024, n_heads=8, mlp=2752, vocab=2048, det_token_idx=2047, clip_dim=256, clip_heads=4,
                               clip_mlp=1024, sam_dim=320, sam_heads=4)


_ORACLE = {}


def oracle_inference(dev, which, outliers=0.0):
    """The fp32 CPU oracle's inference forward on the seed-11 input (B=1, T=8, L=128, n_det=3), ONCE per (dims, outliers) and process:
    the bf16 parity case, the fp8 case and the seeds test all compare against the same pass (round 5, VERDICT r4 next #9a: the full-width
    pass costs ~45 s of host time each and the suite sat at 632 s of the driver's 1200 s)."""
    key = (which, outliers)
    if key in _ORACLE:
        return _ORACLE[key]
    from grove_amd.synthetic import FULL, synthetic_batch
    from oracle import grove_oracle as O
    d = FULL if which == "full" else deep_narrow_dims()
    kw = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=11).as_kwargs(inference=True)
    sd = LazyRoundedWeights(d, gen_device=dev, outliers=outliers)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    gi, si = kw["global_enc_images"].to(bf).float(), kw["grounding_enc_images"].to(bf).float()
    t0 = time.time()
    with torch.no_grad():
        emb_o = O.sam_image_encoder(sd, d, si)
        feats_o, hs_o = O.encode_images(sd, d, gi)
        embeds, _, _ = O.splice(sd, kw["input_ids"], None, None, feats_o)
        hidden_o = O.llama_forward(sd, d, embeds, None)
        pemb = O.pred_embeddings(sd, d, hidden_o, O.det_token_mask(d, kw["input_ids"]))
        _, _, box_o, obj_o = O.decode_boxes(sd, d, pemb, emb_o, kw["original_size_list"], O.dense_pe(sd, d), True)
        _, _, box_ob, obj_ob = O.decode_boxes(sd, d, pemb, emb_o, kw["original_size_list"], O.dense_pe(sd, d, dtype=bf).float(), True)
    _ORACLE[key] = {"emb_o": emb_o, "feats_o": feats_o, "hs_m2": hs_o[-1], "hidden_o": hidden_o, "box_o": box_o, "obj_o": obj_o, "box_ob": box_ob,
                    "obj_ob": obj_ob, "seconds": time.time() - t0}
    return _ORACLE[key]


def run_inference_parity(dev, which, outliers=0.0):
    from grove_amd import GROVEForCausalLM
    from grove_amd.model.decoder import BoxDecoder
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    d = FULL if which == "full" else deep_narrow_dims()
    t0 = time.time()
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf, outliers=outliers)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8)  # pe_dtype: bf16 default
    del sd_dev
    torch.cuda.empty_cache()
    batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=11)
    kw = batch.as_kwargs(inference=True)
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    out_bf = model(**kd)                                   # product default: dense PE in bf16 (quirk Q10)
    model.decoder = BoxDecoder(model._sd, d, dev, grads=model._grad, pe_dtype=torch.float32)
    out = model(**kd)                                      # fp32 PE: the arithmetic-parity configuration of the other tests
    feats_h, clip_h = model(mode="encode_images", images=kd["global_enc_images"])
    torch.cuda.synchronize()
    t_gpu = time.time() - t0

    orc = oracle_inference(dev, which, outliers)
    emb_o, feats_o, hidden_o, box_o, obj_o, box_ob, obj_ob = (orc[k] for k in ("emb_o", "feats_o", "hidden_o", "box_o", "obj_o", "box_ob", "obj_ob"))
    hs_o = [orc["hs_m2"]]
    t_cpu = orc["seconds"]

    g = d.sam_grid
    emb_h = out["image_embeddings"].float().cpu().view(8, g, g, -1).permute(0, 3, 1, 2)
    res = {
        "config": ("FULL dims (LLaMA 32x4096, CLIP 24x1024, SAM 32x1280)" if which == "full" else
                   "full depth at quarter width (LLaMA 32x1024, CLIP 24x256, SAM 32x320)") + ", B=1, T=8, L=128, n_det=3, inference forward, synthetic weights",
        "instances": int(box_o.shape[0]),
        "box_l1_vs_oracle_full": (out["flat_boxes"].cpu() - box_o).abs().mean().item(),
        "box_l1_max_full": (out["flat_boxes"].cpu() - box_o).abs().max().item(),
        "box_l1_bf16_pe_vs_oracle_bf16_pe": (out_bf["flat_boxes"].cpu() - box_ob).abs().mean().item(),
        "objectness_logit_abs_err": (out["flat_logits"].cpu() - obj_o).abs().max().item(),
        "objectness_logit_abs_err_bf16_pe": (out_bf["flat_logits"].cpu() - obj_ob).abs().max().item(),
        "llama_hidden_rel_max": rel(out["hidden"], hidden_o), "llama_hidden_rel_rms": rel_rms(out["hidden"], hidden_o),
        "clip_hidden_m2_rel_max": rel(clip_h.hidden_states[-1], hs_o[-1]), "clip_hidden_m2_rel_rms": rel_rms(clip_h.hidden_states[-1], hs_o[-1]),
        "projected_features_rel_max": rel(feats_h, feats_o),
        "sam_embeddings_rel_max": rel(emb_h, emb_o), "sam_embeddings_rel_rms": rel_rms(emb_h, emb_o),
        "oracle_cpu_seconds": round(t_cpu, 1), "cpu_threads": torch.get_num_threads(), "gpu_seconds_incl_build": round(t_gpu, 1),
        "targets": {"box_l1": 1e-3, "objectness": 5e-2, "hidden_rel": 3e-2},
    }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    res["outliers"] = outliers
    res["llama_stream_abs_max_oracle"] = float(hidden_o.abs().max())
    with open(os.path.join(ROOT, "gpurun_out", f"full_depth_parity_{which}{tag(outliers)}.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    return res


def test_batched_clips_are_batch_invariant_full_width(dev):
    """the clip-batched `infer_clips_batched` must return a clip's ids BY CONSTRUCTION, not by luck of the margins.
    FULL WIDTH (LLaMA 4096 x 2 layers, CLIP 1024 x 3, SAM 1280 x 4 blocks with a global one — the GEMM shapes, tile plans and GEMV
    kernels of the real model; depth only repeats them): three different clips decoded together, the same clips one at a time, and
    regrouped — under `model.batch_invariant_mode()` (whole-tile GEMM plans, the matrix-core GEMV and eight K / V splits for every
    number of sequences) every clip's greedy ids, its generated rows' boxes and its objectness logits are the SAME BITS in all three runs."""
    import dataclasses
    from grove_amd import GROVEForCausalLM
    from grove_amd.infer import infer_clips_batched
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    d = dataclasses.replace(FULL, n_layers=2, clip_layers=4, sam_depth=4, sam_global=(1, 3))
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    sd_dev["model.embed_tokens.weight"] = sd_dev["model.embed_tokens.weight"] * 64.0  # (a stream that walks: see test_full_size_greedy_ids_vs_oracle)
    for k in ("model.mm_projector.2.weight", "model.mm_projector.2.bias"):           # ... and visual tokens loud enough to steer it
        sd_dev[k] = sd_dev[k] * 256.0
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8)
    del sd_dev
    clips = []
    for seed in (31, 32, 33):
        b = synthetic_batch(d, B=1, T=16, L=24, n_det=2, seed=seed)
        clips.append((b.global_enc_images.to(bf), b.grounding_enc_images.to(bf), b.original_size_list[0]))
    prompt = synthetic_batch(d, B=1, T=16, L=24, n_det=2, seed=31).input_ids[0, :20].clone()
    new = 12
    together = infer_clips_batched(model, clips, prompt, max_tokens_new=new)
    alone = [infer_clips_batched(model, [c], prompt, max_tokens_new=new)[0] for c in clips]
    regrouped = infer_clips_batched(model, [clips[2], clips[0]], prompt, max_tokens_new=new)
    ids = [r["output_ids"] for r in together]
    assert all(len(set(i.tolist()[-new:])) >= 4 for i in ids), ids  # walks, not one id repeated
    print("distinct id rows among the three clips:", len({tuple(i.tolist()) for i in ids}))
    for n, (a, b_) in enumerate(list(zip(together, alone)) + [(together[2], regrouped[0]), (together[0], regrouped[1])]):
        assert torch.equal(a["output_ids"], b_["output_ids"]), (n, a["output_ids"], b_["output_ids"])
        for f in range(len(a["pred_bboxes"])):
            assert torch.equal(a["pred_bboxes"][f].cpu(), b_["pred_bboxes"][f].cpu()), (n, f)
            la, lb = a["logits_temp_objectness"][f], b_["logits_temp_objectness"][f]
            assert (la is None and lb is None) or torch.equal(la.cpu(), lb.cpu()), (n, f)
    # the ids of these random-weight clips coincide (the visual tokens do not steer an untrained decoder), so the strong form of the
    # check is on the tensors behind them: visual tokens, SAM embeddings, the prefill's hidden states and every decode step's hidden
    # row, three clips together against each alone — bit for bit, and different from clip to clip
    with model.batch_invariant_mode():
        g3 = torch.cat([c[0][:, :, :8] for c in clips], 0).contiguous().to(dev)
        s3 = torch.cat([c[1][:, :, :8] for c in clips], 0).contiguous().to(dev)
        f3, _ = model(mode="encode_images", images=g3)
        e3 = model(mode="get_grounding_encoder_embs", images=s3)
        o3 = model.generate(input_ids=prompt[None].repeat(3, 1).to(dev), image_features=f3, max_new_tokens=new, eos_token_id=-1,
                            output_hidden_states=True, return_dict_in_generate=True)
        h3 = torch.cat(o3.hidden_states, 1)
        assert not torch.equal(f3[0], f3[1]) and not torch.equal(h3[0], h3[1]) and not torch.equal(e3[:8], e3[8:16])
        for n in range(3):
            f1, _ = model(mode="encode_images", images=g3[n:n + 1])
            e1 = model(mode="get_grounding_encoder_embs", images=s3[n:n + 1])
            o1 = model.generate(input_ids=prompt[None].to(dev), image_features=f1, max_new_tokens=new, eos_token_id=-1, output_hidden_states=True,
                                return_dict_in_generate=True)
            assert torch.equal(f1[0], f3[n]), f"clip {n}: visual tokens"
            assert torch.equal(e1, e3[8 * n:8 * n + 8]), f"clip {n}: SAM embeddings"
            assert torch.equal(o1.sequences[0], o3.sequences[n]), f"clip {n}: ids"
            assert torch.equal(torch.cat(o1.hidden_states, 1)[0], h3[n]), f"clip {n}: hidden states of the prefill and of every decode step"
    # without the mode the same call is allowed to differ in the last bits (other tile plans, other GEMV kernel) — it must still agree closely
    loose = infer_clips_batched(model, clips, prompt, max_tokens_new=new, batch_invariant=False)
    for a, b_ in zip(together, loose):
        for f in range(len(a["pred_bboxes"])):
            if a["pred_bboxes"][f].shape == b_["pred_bboxes"][f].shape and a["pred_bboxes"][f].numel():
                assert (a["pred_bboxes"][f].float().cpu() - b_["pred_bboxes"][f].float().cpu()).abs().max().item() / 640 < 5e-3


def test_full_depth_with_active_clip_adapters(dev):
    """A TRAINED checkpoint's CLIP adapters are active (alpha != 0; SURVEY's synthetic weights have them at 0 and the conv is skipped):
    full depth at quarter width with all eight at alpha = 0.1 — 23 layers, 8 Conv3d adapters on the 16 x 36 grid behind the CLS row —
    against the fp32 oracle, through the Winograd form (round 6: 576 tiles per 8-frame group padded to 768 per transform point) and
    through the 27-tap implicit GEMM: the visual tokens, the LLaMA state they feed and the boxes."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = deep_narrow_dims()
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    names = [k for k in sd_dev if "vision_tower" in k and k.endswith(".alpha")]
    assert len(names) == d.clip_layers // 3
    for k in names:
        sd_dev[k] = torch.full_like(sd_dev[k], 0.1)
    sd = {k: v.float().cpu() for k, v in sd_dev.items()}
    batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=11)
    kw = batch.as_kwargs(inference=True)
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
        kw[k] = kw[k].to(bf).float()
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        ref = O.model_forward(sd, d, **kw)
        feats_o, _ = O.encode_images(sd, d, kw["global_enc_images"])
        feats_0, _ = O.encode_images({**sd, **{k: torch.zeros_like(sd[k]) for k in names}}, d, kw["global_enc_images"])
    assert rel_rms(feats_0, feats_o) > 2e-2, "the adapters must move the visual tokens far beyond the tolerance"
    res = {}
    for form in ("winograd", "direct"):
        model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
        assert all(A["active"] for A in model.clip.adapters) and model.clip.wino
        model.clip.wino = form == "winograd"
        out = model(**kd)
        feats_h, _ = model(mode="encode_images", images=kd["global_enc_images"])
        res[form] = {"box_l1": (out["flat_boxes"].cpu() - ref["flat_boxes"]).abs().mean().item(), "features_rel_rms": rel_rms(feats_h, feats_o),
                     "hidden_rel_rms": rel_rms(out["hidden"], ref["hidden"])}
        del model
    with open(os.path.join(ROOT, "gpurun_out", "full_depth_active_clip_adapters_deep_narrow.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    for form, r in res.items():
        assert r["features_rel_rms"] < 1e-2 and r["hidden_rel_rms"] < 1.5e-2 and r["box_l1"] < 1.5e-3, (form, r)
    assert res["winograd"]["features_rel_rms"] < 1.5 * res["direct"]["features_rel_rms"] + 1e-3, res


@pytest.mark.parametrize("which", ["deep_narrow", "full"])
def test_full_depth_inference_vs_fp32_oracle(dev, which):
    res = run_inference_parity(dev, which)
    assert res["box_l1_vs_oracle_full"] <= 1e-3, res
    assert res["box_l1_bf16_pe_vs_oracle_bf16_pe"] <= 1e-3, res
    assert res["objectness_logit_abs_err"] <= 5e-2 and res["objectness_logit_abs_err_bf16_pe"] <= 5e-2, res
    assert res["llama_hidden_rel_max"] <= 3e-2 and res["clip_hidden_m2_rel_max"] <= 3e-2 and res["sam_embeddings_rel_max"] <= 3e-2, res


GROUPS = (("embed_tokens", "model.embed_tokens."), ("lm_head", "lm_head."), ("mm_projector", "model.mm_projector."),
          ("text_hidden_fcs", "model.text_hidden_fcs."), ("box_decoder", "model.grounding_encoder.mask_decoder."),
          ("sam_adapter_0", "model.grounding_encoder.image_encoder.adapters.0."), ("sam_adapter_1", "model.grounding_encoder.image_encoder.adapters.1."),
          ("sam_adapter_2", "model.grounding_encoder.image_encoder.adapters.2."), ("sam_adapter_3", "model.grounding_encoder.image_encoder.adapters.3."))


def run_training_parity(dev, which, outliers=0.0, stream_dtype=None, seed=11):
    """the configuration the headline bench times — a `train=True` model (bf16 residual streams, bf16 box
    decoder, tape + saved activations) — at FULL DEPTH: 32 LLaMA layers of dgrad, 24 SAM blocks of dgrad, 4 Conv3d adapters with
    weight gradients, against torch autograd through the fp32 CPU oracle on the same bf16-rounded weights: the five loss terms
    within 1 %, cosine > 0.98 and norm ratio in [0.9, 1.1] for every trainable gradient GROUP (the per-tensor figures go to
    gpurun_out/full_depth_training_parity_<which>.json). `which = full` (full width: ~50 GB of host memory for the oracle's
    autograd graph) runs only with GROVE_FULL_TRAIN_PARITY=1; its figures are committed under profiles/."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = FULL if which == "full" else deep_narrow_dims()
    names = trainable_names(d)
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf, outliers=outliers)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, train=True,
                             stream_dtype=stream_dtype)
    batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=seed)
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
    torch.cuda.synchronize()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    t0 = time.time()
    sdg = {k: v.float().cpu().requires_grad_(k in names) for k, v in sd_dev.items()}
    del sd_dev
    ref = O.model_forward(sdg, d, **kw)
    # diagnostic: how much of a gradient-norm difference on the box path is the LOSS SURFACE at two different forward results (GIoU's
    # gradient is piecewise smooth: an edge of the enclosing / intersection box changing hands moves it discontinuously), not a backward
    # defect? d(giou + l1) / d boxes at the oracle's own boxes vs at the HIP path's boxes, both by torch autograd.
    g0 = torch.autograd.grad(ref["giou_loss"] + ref["l1_loss"], ref["flat_boxes"], retain_graph=True)[0]
    hb = out["flat_boxes"].detach().float().cpu().requires_grad_(True)
    nd = hb.shape[0] // 8
    lc = O.loss_components(torch.zeros(()), [[hb[t * nd:(t + 1) * nd] for t in range(8)]],
                           [[ref["flat_logits"].detach()[t * nd:(t + 1) * nd] for t in range(8)]], kw["bboxes_list"], kw["temp_objectness_labels_list"])
    g1 = torch.autograd.grad(lc["giou_loss"] + lc["l1_loss"], hb)[0]
    box_grad_surface = {"norm_ratio_at_hip_boxes_vs_oracle_boxes": float(g1.norm() / g0.norm()),
                        "cos": float(torch.nn.functional.cosine_similarity(g1.flatten(), g0.flatten(), dim=0))}
    ref["loss"].backward()
    t_cpu = time.time() - t0
    terms = ("loss", "ce_loss", "giou_loss", "l1_loss", "temp_objectness_loss")
    loss_rel = {k: abs(float(out[k]) - float(ref[k])) / max(abs(float(ref[k])), 1e-12) for k in terms}
    per_tensor, groups = {}, {}
    gs = {g: ([], []) for g, _ in GROUPS}
    for n in names:
        g, r = model._grad[n].detach().float().cpu(), sdg[n].grad
        if n.endswith("conv3d.weight"):
            g = g.view(r.shape[0], 3, 3, 3, r.shape[1]).permute(0, 4, 1, 2, 3)
        g, r = g.reshape(-1), r.reshape(-1)
        c64, nr64 = cos_norm64(g, r)
        per_tensor[n] = {"cos": c64 if float(r.norm()) > 1e-9 else None, "norm_ratio": nr64, "ref_norm": float(r.double().norm())}
        for gname, pre in GROUPS:
            if n.startswith(pre):
                gs[gname][0].append(g)
                gs[gname][1].append(r)
    for gname, (a, b) in gs.items():
        a, b = torch.cat(a), torch.cat(b)
        c64, nr64 = cos_norm64(a, b)
        groups[gname] = {"cos": c64, "norm_ratio": nr64, "elements": int(a.numel())}
    allg = torch.cat([torch.cat(a) for a, _ in gs.values()])
    allr = torch.cat([torch.cat(b) for _, b in gs.values()])
    res = {"config": ("FULL dims" if which == "full" else "full depth at quarter width (LLaMA 32x1024, CLIP 24x256, SAM 32x320)") +
           ", train=True model (bf16 streams, bf16 decoder), B=1, T=8, L=128, n_det=3, fwd + bwd vs torch autograd through the fp32 oracle",
           "loss_terms_rel_err": loss_rel, "losses": {k: float(out[k]) for k in terms}, "oracle_losses": {k: float(ref[k]) for k in terms},
           "gradient_groups": groups, "whole_gradient": dict(zip(("cos", "norm_ratio"), cos_norm64(allg, allr)), elements=int(allg.numel()), accumulated_in="fp64"),
           "box_loss_gradient_at_hip_boxes_vs_oracle_boxes": box_grad_surface,
           "box_l1_train_mode_vs_oracle": (out["flat_boxes"].detach().cpu() - ref["flat_boxes"].detach()).abs().mean().item(),
           "objectness_logit_abs_err": (out["flat_logits"].detach().cpu() - ref["flat_logits"].detach()).abs().max().item(),
           "worst_tensors": sorted(((v["cos"], n) for n, v in per_tensor.items() if v["cos"] is not None))[:8],
           "largest_box_path_tensors": [(n, round(v["cos"], 4), round(v["norm_ratio"], 4), v["ref_norm"]) for n, v in
                                        sorted(per_tensor.items(), key=lambda kv: -kv[1]["ref_norm"])
                                        if v["cos"] is not None and ("mask_decoder" in n or "text_hidden_fcs" in n)][:16],
           "oracle_cpu_seconds": round(t_cpu, 1)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    res["outliers"] = outliers
    res["stream_dtype"] = str(stream_dtype) if stream_dtype is not None else "default (bf16 for training models)"
    with open(os.path.join(ROOT, "gpurun_out", f"full_depth_training_parity_{which}{tag(outliers)}{'_f32stream' if stream_dtype == torch.float32 else ''}.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    return res


@pytest.mark.parametrize("which", ["deep_narrow", "full"])
def test_full_depth_training_vs_oracle_autograd(dev, which):
    """[full] = the configuration the headline bench times at its REAL width (VERDICT r3 missing #4: it used to run only under
    GROVE_FULL_TRAIN_PARITY=1, so the driver never saw it). ~100 s of oracle autograd on 64 threads and ~50 GB of host memory: skipped —
    loudly, with the reason — only when the host cannot hold it (GROVE_FULL_TRAIN_PARITY=1 forces it, =0 skips it)."""
    if which == "full":
        force = os.environ.get("GROVE_FULL_TRAIN_PARITY")
        if force == "0" or (force is None and _host_memory_gb() < 100):
            pytest.skip(f"full-width training parity needs ~50 GB of host memory for the oracle's autograd graph ({_host_memory_gb():.0f} GB available)")
    res = run_training_parity(dev, which)
    loss_rel, groups = res["loss_terms_rel_err"], res["gradient_groups"]
    assert max(loss_rel.values()) <= 1e-2, loss_rel
    bad = {g: v for g, v in groups.items() if not (v["cos"] > 0.98 and 0.9 < v["norm_ratio"] < 1.1)}
    assert not bad, bad
    # training models keep the reference's bf16 streams and the bf16 decoder: their boxes sit above the inference
    # models' figure, and ONE seed of it is a noisy sample — any change of a rounding pattern anywhere in the 32 layers redraws it:
    # deep-narrow over 5 batch seeds x 2 builds (profiles/r04_training_box_l1_seeds.json) 0.94e-3 .. 2.35e-3, mean 1.4-1.9e-3; full
    # width 2.56e-3 (round 3) and 1.26e-3 (round 4) on the same seed, 2.78e-3 / 1.82e-3 / 3.20e-3 on seeds 12-14
    # (profiles/r04_full_width_training_parity_seeds.json; losses <= 0.5 %, every gradient group's cosine >= 0.985 there). The bound is
    # 1.5x the largest seen; the losses above are the gate.
    assert res["box_l1_train_mode_vs_oracle"] <= (4.8e-3 if which == "full" else 3.5e-3), res["box_l1_train_mode_vs_oracle"]


def test_full_size_greedy_ids_vs_oracle(dev):
    """VERDICT r2 item 2(b), r5 next #7c: caption token ids at FULL size (LLaMA-7B geometry, CLIP ViT-L), not the HIP path against itself:
    `generate` (prefill + cached GEMV steps from one HIP graph) produces 16 new tokens; the fp32 CPU oracle decodes the same stream with
    ITS OWN key / value cache (oracle.llama_forward_cached — pinned to the uncached form and to the reference's golden ids in
    tests/test_oracle_golden.py; HF greedy = argmax of the last position, GROVE.py:418-422), teacher-forced on the HIP path's tokens so
    that every step compares the same prefix. The token embeddings are 64x louder than the synthetic default (a power of two: exact in
    bf16, applied to both sides): with N(0, 0.02) tables the next id hardly depends on the last one and the stream is one id repeated
    (round 5's four tokens were); now it walks. A token must equal the oracle's argmax unless the oracle's margin over the token the
    HIP path chose is inside twice the measured logit error of that step (a bf16 path cannot resolve a smaller margin)."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    import torch.nn.functional as Fn
    d = FULL
    LOUD = 64.0
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    sd_dev["model.embed_tokens.weight"] = sd_dev["model.embed_tokens.weight"] * LOUD
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8)
    del sd_dev
    torch.cuda.empty_cache()
    batch = synthetic_batch(d, B=1, T=8, L=64, n_det=1, seed=5)
    gi = batch.global_enc_images.to(bf)
    prompt = batch.input_ids[:, :40].contiguous()  # BOS, prompt ids, one -200, prompt ids (un-padded: quirk Q9)
    new = 16
    feats, _ = model(mode="encode_images", images=gi.to(dev))
    seqs = model.generate(input_ids=prompt.to(dev), image_features=feats, max_new_tokens=new, eos_token_id=-1)
    seqs_nc = model.generate(input_ids=prompt.to(dev), image_features=feats, max_new_tokens=new, eos_token_id=-1, use_cache=False)
    seqs = seqs.cpu()
    assert torch.equal(seqs, seqs_nc.cpu()), "cached and uncached HIP streams differ"
    sd = LazyRoundedWeights(d, gen_device=dev, scale={"model.embed_tokens.weight": LOUD}, keep_prefix="model.layers.", keep_fp32=_host_memory_gb() > 60)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    t0 = time.time()
    steps = []
    with torch.no_grad():
        feats_o, _ = O.encode_images(sd, d, gi.float())
        emb_table, lm_head = sd["model.embed_tokens.weight"].clone(), sd["lm_head.weight"].clone()
        cache = []
        embeds, _, _ = O.splice(sd, prompt, None, None, feats_o)
        hidden_o = O.llama_forward_cached(sd, d, embeds, cache)
        for t in range(new):
            logits_o = Fn.linear(hidden_o[:, -1], lm_head)[0]
            prefix = seqs[:, :40 + t]
            lh = model.lm_forward(input_ids=prefix.to(dev), image_features=feats, use_cache=False, last_logits_only=True).logits.float().cpu().reshape(-1)[:d.vocab]
            err = (lh - logits_o).abs().max().item()
            top2 = logits_o.topk(2)
            tok = int(seqs[0, 40 + t])
            steps.append({"step": t, "hip_token": tok, "oracle_argmax": int(top2.indices[0]), "oracle_top2_gap": float(top2.values[0] - top2.values[1]),
                          "oracle_margin_over_hip_token": float(top2.values[0] - logits_o[tok]), "logit_abs_err": err,
                          "logit_rel_rms": float((lh - logits_o).pow(2).mean().sqrt() / logits_o.pow(2).mean().sqrt())})
            if t + 1 < new:  # the HIP path's token goes in: both sides always continue the same prefix
                hidden_o = O.llama_forward_cached(sd, d, emb_table[tok][None, None], cache)
    toks = [s_["hip_token"] for s_ in steps]
    res = {"config": f"FULL dims, B=1, T=8, prompt 40 ids (+575 visual), {new} greedy tokens, token embeddings x{LOUD:g}; oracle = fp32 CPU with its own "
                     "K / V cache, teacher-forced on the HIP tokens", "steps": steps, "distinct_tokens": len(set(toks)),
           "ids_equal": all(s_["hip_token"] == s_["oracle_argmax"] for s_ in steps), "oracle_cpu_seconds": round(time.time() - t0, 1)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "full_size_greedy_parity.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    assert res["distinct_tokens"] >= 4, toks  # not one id repeated
    logit_scale = max(1.0, float(logits_o.abs().max()))
    for s_ in steps:
        assert s_["hip_token"] == s_["oracle_argmax"] or s_["oracle_margin_over_hip_token"] <= 2 * s_["logit_abs_err"], s_
        assert s_["logit_abs_err"] <= 0.1 * logit_scale, s_  # a looser path than a few % of the logit scale would be a defect


def test_full_depth_box_l1_over_seeds(dev):
    """How much of the full-depth box figure is the case, not the kernels? The box metric amplifies last-bit differences of the
    text path (re-ordering two fp32 partial sums of one GEMM moved it by 1.3e-4 in round 2: the stream-K fix-up adds a tile's K ranges
    in K order, deterministically, but not in the order of an un-split accumulation chain — both are valid fp32 sums). So the figure
    is reported as a distribution: the deep-narrow model (full depth, quarter width) on four input seeds; mean <= 1e-3 is the gate."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = deep_narrow_dims()
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    sd = {k: v.float().cpu() for k, v in sd_dev.items()}
    del sd_dev
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    vals, maxs = [], []
    for seed in (21, 22, 23, 24):
        batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=seed)
        kw = batch.as_kwargs(inference=True)
        kd = dict(kw)
        for k in ("global_enc_images", "grounding_enc_images"):
            kd[k] = kw[k].to(dev).to(bf)
            kw[k] = kw[k].to(bf).float()
        for k in ("input_ids", "labels", "attention_masks", "offset"):
            kd[k] = kw[k].to(dev)
        out = model(**kd)
        with torch.no_grad():
            ref = O.model_forward(sd, d, **kw)
        e = (out["flat_boxes"].cpu() - ref["flat_boxes"]).abs()
        vals.append(e.mean().item())
        maxs.append(e.max().item())
    res = {"config": "full depth at quarter width, inference model, B=1, T=8, L=128, n_det=3, seeds 21-24", "box_l1": vals, "box_l1_max": maxs,
           "mean": sum(vals) / len(vals)}
    with open(os.path.join(ROOT, "gpurun_out", "full_depth_box_l1_seeds.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    assert res["mean"] <= 1e-3 and max(vals) <= 1.3e-3, res


def _host_memory_gb():
    """min(MemAvailable, what the cgroup still allows) in GB."""
    avail = 0.0
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
                avail = min(avail, (int(m) - int(open(cur).read().strip())) / 1e9)
        except (OSError, ValueError):
            pass
    return avail


def test_full_width_box_l1_over_seeds(dev):
    """the inference box L1 at full depth AND full width was one input (seed 11: 7.9e-4 mean, 2.1e-3 max
    against the 1e-3 north-star bound — 21 % headroom on a metric that moved 1.3e-4 on a sum-order change). Here: the real model
    (LLaMA 32 x 4096, CLIP 24 x 1024, SAM 32 x 1280) on three more inputs, the assert on the MEAN; every figure is written to
    gpurun_out/full_width_box_l1_seeds.json before the assertion."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = FULL
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    # round 6: the DELIVERED fp8 configuration — gemm_dtype="fp8" with its default policy "sam_mlp" (SAM mlp.lin1 / lin2
    # in e4m3, CLIP and LLaMA bf16) — on the same three inputs against the same oracle passes: the 1e-3 bar of the bf16 path, not a bound of its own
    model8 = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, gemm_dtype="fp8")
    assert model8.fp8_policy == "sam_mlp" and all("w1_q" in B and "w2_q" in B for B in model8.sam.blocks) and not model8.llama.fp8
    del sd_dev
    torch.cuda.empty_cache()
    sd = LazyRoundedWeights(d, gen_device=dev)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    vals, maxs, objs, secs = [], [], [], []
    vals8, maxs8, objs8 = [], [], []
    for seed in (11, 21, 22):  # (seed 11's oracle pass is the cached one of the parity case above: two fresh ~45 s passes instead of three)
        batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=seed)
        kw = batch.as_kwargs(inference=True)
        kd = dict(kw)
        for k in ("global_enc_images", "grounding_enc_images"):
            kd[k] = kw[k].to(dev).to(bf)
            kw[k] = kw[k].to(bf).float()
        for k in ("input_ids", "labels", "attention_masks", "offset"):
            kd[k] = kw[k].to(dev)
        out = model(**kd)
        out8 = model8(**kd)
        t0 = time.time()
        if seed == 11:
            orc = oracle_inference(dev, "full")
            ref = {"flat_boxes": orc["box_o"], "flat_logits": orc["obj_o"]}
        else:
            with torch.no_grad():
                ref = O.model_forward(sd, d, **kw)
        secs.append(round(time.time() - t0, 1))
        e = (out["flat_boxes"].cpu() - ref["flat_boxes"]).abs()
        vals.append(e.mean().item())
        maxs.append(e.max().item())
        objs.append((out["flat_logits"].cpu() - ref["flat_logits"]).abs().max().item())
        e8 = (out8["flat_boxes"].cpu() - ref["flat_boxes"]).abs()
        vals8.append(e8.mean().item())
        maxs8.append(e8.max().item())
        objs8.append((out8["flat_logits"].cpu() - ref["flat_logits"]).abs().max().item())
    res = {"config": "FULL dims (LLaMA 32x4096, CLIP 24x1024, SAM 32x1280), inference model (fp32 streams, fp32 box path), B=1, T=8, L=128, n_det=3, seeds 11, 21, 22",
           "box_l1": vals, "box_l1_max": maxs, "objectness_logit_abs_err": objs, "mean": sum(vals) / len(vals), "oracle_seconds": secs,
           "fp8_sam_mlp": {"box_l1": vals8, "box_l1_max": maxs8, "objectness_logit_abs_err": objs8, "mean": sum(vals8) / len(vals8),
                           "what": "gemm_dtype='fp8', fp8_policy='sam_mlp' (the fp8 default): SAM mlp.lin1 / lin2 on grove_gemm_fp8, everything else as the bf16 model"},
           "seed_11_recorded": "profiles/r03_full_depth_parity_full.json: 7.9e-4 mean / 2.1e-3 max"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "full_width_box_l1_seeds.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    assert res["mean"] <= 1e-3 and max(objs) <= 5e-2, res
    assert res["fp8_sam_mlp"]["mean"] <= 1e-3 and max(objs8) <= 5e-2, res["fp8_sam_mlp"]


# measured x 1.5 ("the assert at 1.5x measured, not 2x"); the figures and the precision-policy table are in DESIGN.md section 8
# measured (profiles/r03_full_depth_fp8_parity_*.json): deep_narrow all 1.06e-2 / 9.6 %, deep_narrow det16_kv16 6.5e-3 / 8.7 %,
# full det16_kv16 1.29e-2 / 11.4 % (full "all": 1.65e-2 / 12.2 %, profiles/r03_full_depth_fp8_parity_full_all8.json)
FP8_BOUNDS = {("deep_narrow", "all"): {"box_l1": 1.6e-2, "hidden_rms": 0.145}, ("deep_narrow", "det16_kv16"): {"box_l1": 1.0e-2, "hidden_rms": 0.13},
              ("full", "det16_kv16"): {"box_l1": 1.95e-2, "hidden_rms": 0.17},
              # round 4 (profiles/r04_fp8_clip_policy_deep_narrow.json): CLIP tower in bf16 — 4.6e-3 / 6.2 %; with outliers 9.6e-4 (1.84e-2 with CLIP in e4m3)
              ("deep_narrow", "det16_kv16_clip16"): {"box_l1": 7e-3, "hidden_rms": 0.095},
              # FULL dims, the default policy since round 4 (profiles/r04_outlier_stress_full_dims.json): 5.98e-3 / 7.7 % (det16_kv16 there: 1.34e-2);
              # with massive-activation channels 5.0e-4 (det16_kv16: 7.9e-3)
              ("full", "det16_kv16_clip16"): {"box_l1": 9e-3, "hidden_rms": 0.115}}


def run_fp8_parity(dev, which, policy, outliers=0.0):
    """BASELINE config 5's arithmetic at full depth: `gemm_dtype="fp8"` (every CLIP / LLaMA linear layer with K % 128 == 0 on the e4m3
    MFMA GEMM) against the fp32 oracle. e4m3 keeps 3 mantissa bits: ~3.6 % rms rounding per operand, ~5 % per GEMM output on any data
    whose blocks are not dominated by outliers — scaling granularity does not change that (tools/fp8_policy_study.py: per-32 block
    scales 1.01e-2 vs per-row 0.95e-2 box L1 on this model), so this configuration has its OWN tolerance, stated here and in DESIGN
    section 8: it does not meet the 1e-3 of the bf16 path, and is fenced as such."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = FULL if which == "full" else deep_narrow_dims()
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf, outliers=outliers)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, gemm_dtype="fp8",
                             fp8_policy=policy)
    n_q = sum(1 for L in model.llama.layers for k in ("wqkv_q", "wo_q", "wgu_q", "wd_q") if k in L)
    del sd_dev
    torch.cuda.empty_cache()
    batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=11)
    kw = batch.as_kwargs(inference=True)
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    out = model(**kd)
    feats_h, _ = model(mode="encode_images", images=kd["global_enc_images"])
    torch.cuda.synchronize()
    orc = oracle_inference(dev, which, outliers)  # the same seed-11 pass the bf16 case compares against
    feats_o, hidden_o, box_o, obj_o = orc["feats_o"], orc["hidden_o"], orc["box_o"], orc["obj_o"]
    res = {"config": ("FULL dims" if which == "full" else "full depth at quarter width") + ", gemm_dtype=fp8 (CLIP + LLaMA linear layers, e4m3, per-row / "
           "per-output-channel scales), B=1, T=8, L=128, n_det=3, inference forward vs fp32 oracle",
           "fp8_policy": policy, "llama_projections_quantised": n_q, "box_l1_vs_oracle": (out["flat_boxes"].cpu() - box_o).abs().mean().item(),
           "box_l1_max": (out["flat_boxes"].cpu() - box_o).abs().max().item(),
           "objectness_logit_abs_err": (out["flat_logits"].cpu() - obj_o).abs().max().item(),
           "llama_hidden_rel_rms": rel_rms(out["hidden"], hidden_o), "projected_features_rel_rms": rel_rms(feats_h, feats_o),
           "bounds": FP8_BOUNDS.get((which, policy)), "outliers": outliers, "n_q": n_q}
    with open(os.path.join(ROOT, "gpurun_out", f"full_depth_fp8_parity_{which}_{policy}{tag(outliers)}.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    return res


@pytest.mark.parametrize("which,policy", [("deep_narrow", "all"), ("deep_narrow", "det16_kv16"), ("deep_narrow", "det16_kv16_clip16"), ("full", "det16_kv16_clip16")])
def test_full_depth_fp8_inference_vs_fp32_oracle(dev, which, policy):
    res = run_fp8_parity(dev, which, policy)
    n_q = res["n_q"]
    assert n_q > 0
    assert res["box_l1_vs_oracle"] <= FP8_BOUNDS[(which, policy)]["box_l1"] and res["llama_hidden_rel_rms"] <= FP8_BOUNDS[(which, policy)]["hidden_rms"], res


def decode_precision_probe(dev):
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = deep_narrow_dims()
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    sd = {k: v.float().cpu() for k, v in sd_dev.items()}
    batch = synthetic_batch(d, B=1, T=8, L=64, n_det=1, seed=5)
    gi, si = batch.global_enc_images.to(bf), batch.grounding_enc_images.to(bf)
    prompt = batch.input_ids[:, :40].contiguous()
    feats, _ = model(mode="encode_images", images=gi.to(dev))
    new = 9
    out = model.generate(input_ids=prompt.to(dev), image_features=feats, max_new_tokens=new, eos_token_id=-1, output_hidden_states=True,
                         return_dict_in_generate=True)
    ids = out.sequences.cpu()
    hid = torch.cat(out.hidden_states, 1).float().cpu()        # [1, S0 + new - 1, H]
    hid32 = torch.cat(out.hidden_states_f32, 1).cpu() if getattr(out, "hidden_states_f32", None) is not None else None
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        feats_o, _ = O.encode_images(sd, d, gi.float())
        embeds, _, _ = O.splice(sd, ids[:, :-1], None, None, feats_o)
        hid_o = O.llama_forward(sd, d, embeds, None)
        emb_o = O.sam_image_encoder(sd, d, si.float())
        pe = O.dense_pe(sd, d)
    S0 = 575 + 40

    def rms(a, b):
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())

    def boxes(h_rows):  # rows [n, H] -> boxes for 8 frames each through the oracle's box path
        with torch.no_grad():
            te = O.text_hidden_fcs(sd, h_rows)
            emb_list = [te for _ in range(8)]
            _, _, box, obj = O.decode_boxes(sd, d, emb_list, emb_o, [(640, 360)], pe, True)
        return box
    n = hid.shape[1] - S0
    pre_rows = slice(S0 - n, S0)
    res = {"prefill_rows_hidden_rel_rms": rms(hid[0, pre_rows], hid_o[0, pre_rows]), "decode_rows_hidden_rel_rms": rms(hid[0, S0:], hid_o[0, S0:]),
           "box_l1_from_prefill_rows": float((boxes(hid[0, pre_rows]) - boxes(hid_o[0, pre_rows])).abs().mean()),
           "box_l1_from_decode_rows": float((boxes(hid[0, S0:]) - boxes(hid_o[0, S0:])).abs().mean()), "rows": n}
    if hid32 is not None:  # what evaluate()'s box path reads since round 3: the fp32 hidden rows of the fp32 residual streams
        res.update({"f32_prefill_rows_hidden_rel_rms": rms(hid32[0, pre_rows], hid_o[0, pre_rows]), "f32_decode_rows_hidden_rel_rms": rms(hid32[0, S0:], hid_o[0, S0:]),
                    "box_l1_from_f32_prefill_rows": float((boxes(hid32[0, pre_rows]) - boxes(hid_o[0, pre_rows])).abs().mean()),
                    "box_l1_from_f32_decode_rows": float((boxes(hid32[0, S0:]) - boxes(hid_o[0, S0:])).abs().mean())})
    return res




def test_generated_rows_hidden_precision_full_depth(dev):
    """The rows evaluate() decodes boxes from when [DET] tokens are GENERATED come from the cached decode steps (weight-streaming GEMV
    path), not from the prefill. Round 2 kept that step's residual stream in bf16: at full depth the generated rows' hidden state sat
    1.18 % rms from the fp32 oracle (prefill rows with their fp32 stream: 0.48 %) and, through the oracle's own exact box path, moved
    the boxes by 1.13e-3 — over the 1e-3 bar that the prefill rows (6.8e-4) meet. Round 3: fp32 residual stream in the decode step of
    inference models + fp32 hidden rows into the box path (generate().hidden_states_f32). Deep-narrow model, 8 generated positions,
    every one treated as a [DET] row (a precision probe)."""
    res = decode_precision_probe(dev)
    with open(os.path.join(ROOT, "gpurun_out", "decode_rows_precision_deep_narrow.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    assert res["f32_decode_rows_hidden_rel_rms"] < 8e-3 and res["decode_rows_hidden_rel_rms"] < 8e-3, res   # measured 3.7e-3 / 4.1e-3
    assert res["box_l1_from_f32_decode_rows"] < 1e-3 and res["box_l1_from_f32_prefill_rows"] < 1e-3, res    # measured 3.2e-4 / 6.2e-4


def test_outlier_stress_deep_narrow(dev):
    """every precision figure so far came from N(0, 0.02)-like weights — no activation outliers. With
    `outliers=1000` three hidden channels of the LLaMA residual stream carry values ~200x the typical magnitude at every position
    (max ~2500; what LLaMA-7B's massive-activation channels do). Measured (profiles/r04_outlier_stress_deep_narrow.json,
    r04_outlier_training_probe.json; baseline in brackets): inference box L1 7.5e-4 [4.7e-4], hidden 0.76 % rms [0.52 %]; training-mode
    loss terms within 0.24 % [0.04 %], whole gradient cosine 0.993 [0.9985] with the bottom-of-stack groups' NORM off by up to 16 %
    (embed_tokens 0.84, mm_projector 0.88 — and 1.09 / 1.16 with fp32 forward streams: bf16 rounding of a 1000:1 dynamic range along
    32 layers of dgrad, not a sign of a wrong kernel: every group's cosine stays >= 0.989); fp8 det16_kv16 box L1 1.9e-2 [6.5e-3].
    Asserts at ~1.5x the measured values (gradient cosines: below the smallest of five builds)."""
    F = 1000.0
    r = run_inference_parity(dev, "deep_narrow", outliers=F)
    assert r["llama_stream_abs_max_oracle"] > 20.0        # the outliers really are in the stream (5.0 without)
    assert r["box_l1_vs_oracle_full"] <= 1.1e-3 and r["llama_hidden_rel_rms"] <= 1.2e-2 and r["objectness_logit_abs_err"] <= 5e-2, r
    t = run_training_parity(dev, "deep_narrow", outliers=F)
    assert max(t["loss_terms_rel_err"].values()) <= 5e-3, t["loss_terms_rel_err"]
    # (one seed is a sample here too: five builds of round 4 gave whole-gradient cosines 0.983 .. 0.994 and mm_projector 0.9706 .. 0.989
    # on this seed — tools/parity_ab.py, profiles/r04_training_box_l1_seeds.json)
    assert t["whole_gradient"]["cos"] >= 0.97, t["whole_gradient"]
    bad = {g: v for g, v in t["gradient_groups"].items() if not (v["cos"] > 0.95 and 0.78 < v["norm_ratio"] < 1.2)}
    assert not bad, bad
    q = run_fp8_parity(dev, "deep_narrow", "det16_kv16", outliers=F)
    assert q["box_l1_vs_oracle"] <= 2.9e-2 and q["llama_hidden_rel_rms"] <= 0.16, q
    # the excess is the e4m3 CLIP tower (tools/fp8_policy_study.py --outliers: SmoothQuant scaling, bf16 side products for the outlier
    # channels and bf16 first layers change nothing; CLIP in bf16 does): 9.6e-4 measured (profiles/r04_fp8_clip_policy_deep_narrow.json)
    q16 = run_fp8_parity(dev, "deep_narrow", "det16_kv16_clip16", outliers=F)
    assert q16["box_l1_vs_oracle"] <= 1.5e-3 and q16["box_l1_vs_oracle"] < 0.2 * q["box_l1_vs_oracle"], (q16, q)


def test_bench_shape_step_equals_its_windows_full_dims(dev):
    """Round 5 (VERDICT r4 next #9e): the shape the headline bench times — B = 2 clips x T = 16 frames = four 8-frame windows, GEMMs at
    M = 2812 / 32768 with their stream-K cuts, temporal tap skipping and the last-layer tail — is otherwise covered only by the
    bit-repeat race screen; every oracle comparison at full dims runs B = 1, T = 8 (M = 703 / 8192: other tile plans). Here the two
    meet on the SAME model: a batch whose four windows are identical (both clips equal, frames 8..15 = frames 0..7) must give the
    loss terms of the one-window step (every normaliser — labelled tokens, ground-truth boxes, instances — scales with the window
    count) and the same gradient (mean over four equal windows), up to what a different fp32 sum order does to a bf16-stream model:
    the CE path is insensitive to it (measured: CE 2.8e-4 apart, lm_head's gradient at cosine 0.99992, norm ratio 0.99998 — asserted at
    1e-3 / 0.9995 / 0.5 %), the box path is not — two builds of the SAME shape that differ in one GEMM's K cut move the training-mode
    boxes by 1-2e-3 (profiles/r04_training_box_l1_seeds.json) and GIoU's gradient is piecewise smooth — measured here: box
    losses 0.3-0.45 % apart, every box-downstream group at cosine 0.996-0.999 with norm ratio 1.029-1.039, the 481 M-element
    gradient at 0.9975 / 1.034: the same figures the full-width step shows against the ORACLE (cosine >= 0.9925, ratios 1.034-1.048),
    i.e. the bench shape is as far from the one-window shape as either is from the oracle. Asserted: loss terms within 1 %, groups at
    cosine >= 0.99 and norm ratio within 7 %. A wrong normaliser, a dropped window or a wrong tap range shows up as 25-50 %."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict
    d = FULL
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, train=True)
    del sd_dev
    torch.cuda.empty_cache()
    b1 = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=11)
    k1 = b1.as_kwargs()
    one = dict(k1)
    for k in ("global_enc_images", "grounding_enc_images"):
        one[k] = k1[k].to(dev).to(bf)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        one[k] = k1[k].to(dev)
    four = dict(one)
    for k in ("global_enc_images", "grounding_enc_images"):
        four[k] = torch.cat([one[k], one[k]], 2).repeat(2, 1, 1, 1, 1).contiguous()      # [2, 3, 16, H, W]
    for k in ("input_ids", "labels", "attention_masks"):
        four[k] = one[k].repeat(2, 1)
    four["bboxes_list"] = [list(k1["bboxes_list"][0]) * 2 for _ in range(2)]
    four["temp_objectness_labels_list"] = [list(k1["temp_objectness_labels_list"][0]) * 2 for _ in range(2)]
    four["original_size_list"] = [k1["original_size_list"][0]] * 2
    four["offset"] = torch.arange(3, device=dev)
    keys = ("ce_loss", "giou_loss", "l1_loss", "temp_objectness_loss", "loss")
    res = {}
    for name, kw in (("one_window", one), ("bench_shape", four)):
        model.zero_grad()
        out = model(**kw)
        model.backward(out["loss"])
        torch.cuda.synchronize()
        res[name] = ({k: float(out[k]) for k in keys}, model._flat_grad.clone())
    (l1, g1), (l4, g4) = res["one_window"], res["bench_shape"]
    rec = {"losses_one_window": l1, "losses_bench_shape": l4, "groups": {}}
    cos = torch.nn.functional.cosine_similarity(g1, g4, dim=0).item()
    rec["whole_gradient"] = {"cosine": cos, "norm_ratio": (g4.norm() / g1.norm()).item(), "elements": g1.numel()}
    for gname, pre in GROUPS:
        names = [n for n in model.trainable if n.startswith(pre)]
        lo = min(model._grad_off[n] for n in names)
        hi = max(model._grad_off[n] + model._grad_slot[n] for n in names)
        a, c = g1[lo:hi], g4[lo:hi]
        rec["groups"][gname] = {"cosine": torch.nn.functional.cosine_similarity(a, c, dim=0).item(), "norm_ratio": (c.norm() / a.norm().clamp_min(1e-30)).item()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_shape_consistency_full.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))
    assert abs(l4["ce_loss"] - l1["ce_loss"]) <= 1e-3 * abs(l1["ce_loss"]), (l1, l4)
    for k in keys:
        assert abs(l4[k] - l1[k]) <= 1e-2 * max(1.0, abs(l1[k])), (k, l1[k], l4[k])
    lm = rec["groups"]["lm_head"]
    assert lm["cosine"] >= 0.9995 and abs(lm["norm_ratio"] - 1.0) <= 5e-3, lm
    assert cos >= 0.99 and abs(rec["whole_gradient"]["norm_ratio"] - 1.0) <= 0.07, rec["whole_gradient"]
    for gname, v in rec["groups"].items():
        assert v["cosine"] >= 0.99 and abs(v["norm_ratio"] - 1.0) <= 0.07, (gname, v)
