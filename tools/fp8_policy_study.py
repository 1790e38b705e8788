"""Which fp8 precision policy keeps the boxes inside the 1e-3 L1 tolerance? (CPU study, no GPU: fake-quantised oracle)

Every linear layer of the CLIP tower / LLaMA stack that a policy puts in fp8 is evaluated as the HIP path does it: activation
rounded to bf16, quantised per row to e4m3 (amax / 448), weight quantised per output channel, fp32 accumulate, bf16 output.
Policies (LLaMA; CLIP is `clip8` on/off):
  all8        every projection of every layer in fp8                          (round 2's path)
  det16       + the rows of the [DET] positions computed in bf16 (row-selective precision: their q/k/v, o, gate/up, down)
  det16_kv16  + k_proj / v_proj of ALL rows in bf16
  lastN       the last N layers entirely in bf16
  out16       (round 4, with --outliers F) the outlier hidden channels of the norm-following projections (q / k / v / gate / up: their
              input is the RMSNorm of the residual stream) go through a bf16 side product (K = 3 channels), the e4m3 GEMM sees them zeroed
  smooth      SmoothQuant: per-input-channel s_j = sqrt(amax|x_j| / amax|W_j|) folded into the norm weight (x / s) and the weight (W s)
Usage: python tools/fp8_policy_study.py [tiny|deep_narrow] [--outliers F]   (F > 0: synthetic massive activations, grove_amd/synthetic.py)
       python tools/fp8_policy_study.py deep_narrow --sam   (round 5, VERDICT r4 next #10: the SAM arm — e4m3 for the SAM tower's
       mlp.lin1 / lin2 (two thirds of its GEMM FLOPs), then qkv / proj as well, on top of the bf16 path and of the default LLaMA policy)
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bf = torch.bfloat16
E4 = torch.float8_e4m3fn


def q8(x, blk=0):
    """per-row (blk=0) or per-(row, blk-wide block) amax/448 e4m3 fake quantisation of the last dim."""
    x = x.float()
    if blk:
        sh = x.shape
        xb = x.reshape(*sh[:-1], sh[-1] // blk, blk)
        s = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30) / 448.0
        return ((xb / s).to(E4).float() * s).reshape(sh)
    s = x.abs().amax(-1, keepdim=True).clamp_min(1e-30) / 448.0
    return (x / s).to(E4).float() * s


def r16(x):
    return x.to(bf).float()


class Policy:
    def __init__(self, name, clip8=True, llama8=True, det16=False, kv16=False, last_bf16=0, blk=0, first_bf16=0, o16=False, down16=False, det_from=0,
                 out16=False, smooth=False):
        self.__dict__.update(locals())


def lin8(x, w, b, pol):
    y = F.linear(q8(r16(x), pol.blk), q8(w, pol.blk), b)
    return r16(y)


def lin16(x, w, b=None):
    return r16(F.linear(r16(x), w, b))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "tiny"
    outl = float(sys.argv[sys.argv.index("--outliers") + 1]) if "--outliers" in sys.argv else 0.0
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    from tests.test_full_depth_gpu import deep_narrow_dims
    import dataclasses
    if which == "tiny":
        d = dataclasses.replace(TINY, clip_dim=128, clip_heads=2, clip_mlp=256)
        B, L, n_det, seed = 2, 40, 3, 2
    else:
        d = deep_narrow_dims()
        B, L, n_det, seed = 1, 128, 3, 11
    torch.set_num_threads(os.cpu_count() or 8)
    sd = {k: v.to(bf).float() for k, v in synthetic_state_dict(d, outliers=outl).items()}
    from grove_amd.synthetic import outlier_channels
    och = list(outlier_channels(d))
    batch = synthetic_batch(d, B=B, T=8, L=L, n_det=n_det, seed=seed)
    kw = batch.as_kwargs(inference=True)
    gi, si = kw["global_enc_images"].to(bf).float(), kw["grounding_enc_images"].to(bf).float()
    ids = kw["input_ids"]
    with torch.no_grad():
        emb_o = O.sam_image_encoder(sd, d, si)
        feats_o, _ = O.encode_images(sd, d, gi)
        embeds_o, _, _ = O.splice(sd, ids, None, None, feats_o)
        hidden_o = O.llama_forward(sd, d, embeds_o, None)
        mask = O.det_token_mask(d, ids)
        pe = O.dense_pe(sd, d)
        _, _, box_o, obj_o = O.decode_boxes(sd, d, O.pred_embeddings(sd, d, hidden_o, mask), emb_o, kw["original_size_list"], pe, True)
    S = hidden_o.shape[1]
    det_pos = [(575 + (ids[b, 1:] == d.det_token_idx).nonzero().flatten()) for b in range(B)]

    def clip_feats(pol):
        if not pol.clip8:
            f, _ = O.encode_images(sd, d, gi)
            return f
        orig = O._lin

        def lin(sd_, name, x):
            if name.startswith(O.V + "encoder.layers."):
                return lin8(x, sd_[name + ".weight"], sd_.get(name + ".bias"), pol)
            return orig(sd_, name, x)
        O._lin = lin
        try:
            f, _ = O.encode_images(sd, d, gi)
        finally:
            O._lin = orig
        return f

    def llama(pol, embeds):
        Bc, S_, H = embeds.shape
        nh, hd = d.n_heads, d.head_dim
        cos, sin = O._rope_cos_sin(d, torch.arange(S_))
        neg = torch.finfo(torch.float32).min
        add = torch.full((S_, S_), neg).triu(1)[None, None]
        x = embeds

        def proj(name, h, li, kind):
            w = sd[name + ".weight"]
            full16 = (not pol.llama8) or li >= d.n_layers - pol.last_bf16 or li < pol.first_bf16 or (pol.kv16 and kind in ("k", "v")) \
                or (pol.first_bf16 < 0 and li < -pol.first_bf16 and kind in ("o", "down")) \
                or (pol.o16 and kind == "o") or (pol.down16 and kind == "down")
            if full16:
                return lin16(h, w)
            if pol.out16 and kind in ("q", "k", "v", "gate", "up"):
                h0 = h.clone()
                h0[..., och] = 0.0
                y = r16(F.linear(q8(r16(h0), pol.blk), q8(w, pol.blk)) + F.linear(r16(h[..., och]), w[:, och]))
            elif pol.smooth and kind in ("q", "k", "v", "gate", "up"):
                sj = (r16(h).abs().amax(dim=(0, 1)).clamp_min(1e-5) / w.abs().amax(0).clamp_min(1e-5)).sqrt()
                y = r16(F.linear(q8(r16(h / sj), pol.blk), q8(w * sj, pol.blk)))
            else:
                y = lin8(h, w, None, pol)
            if pol.det16 and li >= pol.det_from:
                for b in range(Bc):
                    y[b, det_pos[b]] = lin16(h[b, det_pos[b]], w)
            return y
        for i in range(d.n_layers):
            p = f"model.layers.{i}."
            h = O.rms_norm(x, sd[p + "input_layernorm.weight"], d.rms_eps)
            sh = lambda t: t.view(Bc, S_, nh, hd).transpose(1, 2)  # noqa: E731
            q, k, v = (sh(proj(p + f"self_attn.{n}_proj", h, i, n)) for n in "qkv")
            q = q * cos + O._rotate_half(q) * sin
            k = k * cos + O._rotate_half(k) * sin
            att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + add, -1)
            o = r16((att @ v).transpose(1, 2).reshape(Bc, S_, H))
            x = x + proj(p + "self_attn.o_proj", o, i, "o")
            h = O.rms_norm(x, sd[p + "post_attention_layernorm.weight"], d.rms_eps)
            a = r16(F.silu(proj(p + "mlp.gate_proj", h, i, "gate")) * proj(p + "mlp.up_proj", h, i, "up"))
            x = x + proj(p + "mlp.down_proj", a, i, "down")
        return O.rms_norm(x, sd["model.norm.weight"], d.rms_eps)

    def rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    sam_arm = "--sam" in sys.argv

    def sam_embs(which_layers):
        """SAM tower with the named linear layers of every block fake-quantised to e4m3 (per-row activations, per-channel weights)."""
        if not which_layers:
            return emb_o
        orig = O._lin
        pol0 = Policy("sam")

        def lin(sd_, name, x):
            if name.startswith(O.S + "blocks.") and name.rsplit(".", 2)[-2] + "." + name.rsplit(".", 2)[-1] in which_layers or \
                    (name.startswith(O.S + "blocks.") and name.rsplit(".", 1)[-1] in which_layers):
                return lin8(x, sd_[name + ".weight"], sd_.get(name + ".bias"), pol0)
            return orig(sd_, name, x)
        O._lin = lin
        try:
            return O.sam_image_encoder(sd, d, si)
        finally:
            O._lin = orig

    if sam_arm:
        rows = []
        with torch.no_grad():
            base = Policy("bf16 (no fp8)", clip8=False, llama8=False)
            dflt = Policy("det16_kv16 clip16", det16=True, kv16=True, clip8=False)
            feats = clip_feats(base)
            embeds, _, _ = O.splice(sd, ids, None, None, feats)
            hid = {"bf16": llama(base, embeds), "det16_kv16_clip16": llama(dflt, embeds)}
            for sam_name, layers in (("SAM bf16", ()), ("SAM mlp.lin1 + mlp.lin2 e4m3", ("mlp.lin1", "mlp.lin2")),
                                     ("SAM mlp + qkv + proj e4m3", ("mlp.lin1", "mlp.lin2", "qkv", "proj"))):
                emb = sam_embs(layers)
                for hname, hidden in hid.items():
                    _, _, box, obj = O.decode_boxes(sd, d, O.pred_embeddings(sd, d, hidden, mask), emb, kw["original_size_list"], pe, True)
                    r = {"sam": sam_name, "llama_clip": hname, "sam_emb_rms": rms(emb, emb_o), "box_l1": (box - box_o).abs().mean().item(),
                         "box_max": (box - box_o).abs().max().item(), "obj_abs": (obj - obj_o).abs().max().item()}
                    rows.append(r)
                    print(json.dumps(r), flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/fp8_policy_study_{which}_sam.json", "w") as fh:
            json.dump(rows, fh, indent=1)
        return
    if outl:
        pols = [Policy("bf16 (no fp8)", clip8=False, llama8=False), Policy("all8"), Policy("all8 + out16", out16=True), Policy("all8 + smooth", smooth=True),
                Policy("det16_kv16", det16=True, kv16=True), Policy("det16_kv16 + out16", det16=True, kv16=True, out16=True),
                Policy("det16_kv16 + smooth", det16=True, kv16=True, smooth=True),
                Policy("all8, first 2 layers bf16", first_bf16=2), Policy("det16_kv16, first 2 layers bf16", det16=True, kv16=True, first_bf16=2),
                Policy("det16_kv16, o/down of layers 0-1 bf16", det16=True, kv16=True, first_bf16=-2),
                Policy("det16_kv16 clip16", det16=True, kv16=True, clip8=False), Policy("clip8 llama16", llama8=False)]
    else:
      pols = [Policy("bf16 (no fp8)", clip8=False, llama8=False),
            Policy("all8"),
            Policy("all8 blk32", blk=32),
            Policy("clip16 llama8", clip8=False),
            Policy("clip8 llama16", llama8=False),
            Policy("det16", det16=True),
            Policy("det16 clip16", det16=True, clip8=False),
            Policy("det16_kv16", det16=True, kv16=True),
            Policy("det16_kv16 clip16", det16=True, kv16=True, clip8=False),
            Policy("last2", last_bf16=2), Policy("last8", last_bf16=8),
            Policy("det16 last4", det16=True, last_bf16=4),
            Policy("det16_kv16_o16", det16=True, kv16=True, o16=True),
            Policy("kv16 only", kv16=True),
            Policy("kv16 + det16 from layer 16", det16=True, kv16=True, det_from=16),
            Policy("kv16 + det16 from layer 24", det16=True, kv16=True, det_from=24),
            ]
    rows = []
    with torch.no_grad():
        cache = {}
        for pol in pols:
            ck = (pol.clip8, pol.blk)
            if ck not in cache:
                cache[ck] = clip_feats(pol)
            feats = cache[ck]
            embeds, _, _ = O.splice(sd, ids, None, None, feats)
            hidden = llama(pol, embeds)
            _, _, box, obj = O.decode_boxes(sd, d, O.pred_embeddings(sd, d, hidden, mask), emb_o, kw["original_size_list"], pe, True)
            det_err = torch.cat([hidden[b, det_pos[b]] - hidden_o[b, det_pos[b]] for b in range(B)]).pow(2).mean().sqrt() / \
                torch.cat([hidden_o[b, det_pos[b]] for b in range(B)]).pow(2).mean().sqrt()
            r = {"policy": pol.name, "feat_rms": rms(feats, feats_o), "hidden_rms": rms(hidden, hidden_o), "det_rows_rms": det_err.item(),
                 "box_l1": (box - box_o).abs().mean().item(), "box_max": (box - box_o).abs().max().item(),
                 "obj_abs": (obj - obj_o).abs().max().item()}
            rows.append(r)
            print(json.dumps(r), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/fp8_policy_study_{which}{'_outliers%g' % outl if outl else ''}.json", "w") as fh:
        json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
