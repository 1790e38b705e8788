"""Does a Winograd F(2x2x2, 3x3x3) form of the SAM Conv3d adapters stay inside the parity budget of the direct bf16 form?
(CPU study, no GPU: fake-quantised oracle — VERDICT r5 next #2 (i). image_encoder.py:43-59: Conv3d(C, C, 3, padding=1) on
[B, C, T=8, 32, 32].)

The direct HIP path multiplies the bf16 operands it is given (x = bf16 of the stream, W bf16) with fp32 accumulation. A Winograd
form multiplies TRANSFORMED operands — V = (B^T (x) B^T (x) B^T) x-tile, U = (G (x) G (x) G) w, d M^ = (A (x) A (x) A) dZ-tile —
which have to be rounded to bf16 again before the MFMA: 64 products per 8 outputs instead of 216, two new rounding points per
product. Arms (all with x / dZ rounded to bf16 first, fp32 accumulate, transforms in fp32):
  direct16      what the product does today
  wino_wgrad    forward + dgrad direct, weight gradient = G^T [ sum_tiles dM^ (.) V ] G with V, dM^ rounded to bf16
  wino_all      forward, dgrad and wgrad through the transforms, the forward / dgrad products M^ kept in fp32 until A^T . A
  wino_all_m16  the same with M^ rounded to bf16 (what a GEMM with a bf16 output would store)
Measured against the fp32 oracle (no rounding anywhere) per batch seed: box L1, the four box-path loss terms, cosine / relative rms
error of the adapters' weight / bias / alpha gradients, relative rms of the SAM embeddings. The LLaMA side does not depend on the arm
and is evaluated once per seed. Usage: python tools/winograd_study.py [deep_narrow|tiny] [--seeds 11 12 13 14 15] -> gpurun_out/winograd_study.json
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bf = torch.bfloat16

from oracle.winograd_ref import AT, BT, G, r16, t3, tiles_in, tiles_out, untile_out  # noqa: E402,F401


def wino_conv(x5, w, m16):
    from oracle.winograd_ref import wino_conv as wc
    return wc(x5, w, m16)


def wino_wgrad(V, g5):
    from oracle.winograd_ref import wino_wgrad as ww
    return ww(V, g5)


class AdapterConv(torch.autograd.Function):
    """conv3d(x, w) (no bias) with the arm's arithmetic in forward, dgrad and wgrad."""

    @staticmethod
    def forward(ctx, x5, w, arm):
        x16 = r16(x5)
        ctx.arm = arm
        if arm in ("wino_all", "wino_all_m16"):
            y, V = wino_conv(x16, w, arm == "wino_all_m16")
            ctx.save_for_backward(x16, w, V)
        else:
            y = F.conv3d(x16, w, None, padding=1)
            ctx.save_for_backward(x16, w, None)
        return y

    @staticmethod
    def backward(ctx, g):
        x16, w, V = ctx.saved_tensors
        arm = ctx.arm
        g16 = r16(g)
        if arm == "direct16":
            dw = torch.nn.grad.conv3d_weight(x16, w.shape, g16, padding=1)
        else:
            if V is None:
                V = r16(t3(BT, tiles_in(x16)))
            dw = wino_wgrad(V, g16)
        if arm in ("wino_all", "wino_all_m16"):
            wd = w.flip(2, 3, 4).transpose(0, 1).contiguous()
            dx, _ = wino_conv(g16, wd, arm == "wino_all_m16")
        else:
            dx = torch.nn.grad.conv3d_input(x16.shape, w, g16, padding=1)
        return dx, dw, None


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "deep_narrow"
    seeds = [int(s) for s in sys.argv[sys.argv.index("--seeds") + 1:]] if "--seeds" in sys.argv else [11, 12, 13, 14, 15]
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    from tests.test_full_depth_gpu import deep_narrow_dims
    d = TINY if which == "tiny" else deep_narrow_dims()
    torch.set_num_threads(os.cpu_count() or 8)
    sd0 = {k: v.to(bf).float() for k, v in synthetic_state_dict(d).items()}
    ad_names = [k for k in sd0 if ".image_encoder.adapters." in k]
    arms = ["fp32", "direct16", "wino_wgrad", "wino_all", "wino_all_m16"]
    orig = O.conv_adapter
    state = {"arm": "fp32"}

    def conv_adapter(x5, w, b, alpha):
        if state["arm"] == "fp32":
            return orig(x5, w, b, alpha)
        y = AdapterConv.apply(x5, w, state["arm"]) + b.view(1, -1, 1, 1, 1)
        return torch.tanh(alpha) * F.relu(y) + x5
    out = {"config": which, "seeds": seeds, "arms": arms[1:], "per_seed": {}}
    for seed in seeds:
        batch = synthetic_batch(d, B=1, T=8, L=128 if which != "tiny" else 40, n_det=3, seed=seed)
        kw = batch.as_kwargs()
        gi, si = kw["global_enc_images"].to(bf).float(), kw["grounding_enc_images"].to(bf).float()
        ids = kw["input_ids"]
        with torch.no_grad():  # the arm-independent side, once
            feats, _ = O.encode_images(sd0, d, gi)
            embeds, new_labels, new_mask = O.splice(sd0, ids, kw["labels"], kw["attention_masks"], feats)
            hidden = O.llama_forward(sd0, d, embeds, new_mask)
            emb = O.pred_embeddings(sd0, d, hidden, O.det_token_mask(d, ids))
            pe = O.dense_pe(sd0, d).float()
        res = {}
        for arm in arms:
            t0 = time.time()
            state["arm"] = arm
            sd = {k: (v.clone().requires_grad_(True) if k in ad_names else v) for k, v in sd0.items()}
            O.conv_adapter = conv_adapter
            try:
                ie = O.sam_image_encoder(sd, d, si)
            finally:
                O.conv_adapter = orig
            boxes, logits, flat_box, flat_obj = O.decode_boxes(sd, d, emb, ie, kw["original_size_list"], pe, False)
            lc = O.loss_components(torch.zeros(()), boxes, logits, kw["bboxes_list"], kw["temp_objectness_labels_list"])
            lc["loss"].backward()
            res[arm] = {"boxes": flat_box.detach(), "ie": ie.detach(), "losses": {k: float(v) for k, v in lc.items() if k != "ce_loss"},
                        "grads": {k: sd[k].grad.detach().clone() for k in ad_names}}
            print(f"seed {seed} {arm}: {time.time() - t0:.1f} s", flush=True)
        ref = res["fp32"]
        rows = {}
        for arm in arms[1:]:
            r = res[arm]
            gw = torch.cat([r["grads"][k].flatten() for k in ad_names if k.endswith("conv3d.weight")]).double()
            gw0 = torch.cat([ref["grads"][k].flatten() for k in ad_names if k.endswith("conv3d.weight")]).double()
            small = {}
            for suffix in ("conv3d.bias", "alpha"):
                a = torch.cat([r["grads"][k].flatten() for k in ad_names if k.endswith(suffix)]).double()
                b = torch.cat([ref["grads"][k].flatten() for k in ad_names if k.endswith(suffix)]).double()
                small[suffix] = {"cos": float(F.cosine_similarity(a, b, dim=0)), "rel_rms": float((a - b).norm() / b.norm())}
            rows[arm] = {"box_l1": float((r["boxes"] - ref["boxes"]).abs().mean()),
                         "embeddings_rel_rms": float((r["ie"] - ref["ie"]).norm() / ref["ie"].norm()),
                         "loss_rel": {k: abs(v - ref["losses"][k]) / max(abs(ref["losses"][k]), 1e-12) for k, v in r["losses"].items()},
                         "wgrad": {"cos": float(F.cosine_similarity(gw, gw0, dim=0)), "rel_rms": float((gw - gw0).norm() / gw0.norm()),
                                   "norm_ratio": float(gw.norm() / gw0.norm())},
                         **small}
            print(seed, arm, json.dumps(rows[arm]), flush=True)
        out["per_seed"][seed] = rows
    summ = {}
    for arm in arms[1:]:
        import statistics as st
        v = [out["per_seed"][s][arm] for s in seeds]
        def ms(f):
            xs = [f(x) for x in v]
            return {"mean": st.mean(xs), "stderr": (st.stdev(xs) / len(xs) ** 0.5) if len(xs) > 1 else 0.0}
        summ[arm] = {"box_l1": ms(lambda x: x["box_l1"]), "embeddings_rel_rms": ms(lambda x: x["embeddings_rel_rms"]),
                     "wgrad_rel_rms": ms(lambda x: x["wgrad"]["rel_rms"]), "wgrad_cos": ms(lambda x: x["wgrad"]["cos"]),
                     "loss_rel_max": ms(lambda x: max(x["loss_rel"].values())), "bias_rel_rms": ms(lambda x: x["conv3d.bias"]["rel_rms"]),
                     "alpha_rel_rms": ms(lambda x: x["alpha"]["rel_rms"])}
    out["summary"] = summ
    print(json.dumps(summ, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/winograd_study_{which}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
