"""Where does the box L1 against the fp32 oracle come from? (deep-narrow dims: full depth, quarter width)
  A. HIP decoder fed the ORACLE's tower outputs (SAM embeddings, [DET] text embeddings, rounded to bf16)  -> decoder arithmetic
  B. ORACLE decoder fed the HIP towers' outputs                                                              -> tower error seen through the decoder
  C. end to end.   Usage (GPU box): python tools/box_error_budget.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_full_depth_gpu import LazyRoundedWeights, deep_narrow_dims  # noqa: E402


def main():
    from grove_amd import GROVEForCausalLM, ops
    from grove_amd.model.tape import Var
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    bf = torch.bfloat16
    dev = torch.device("cuda:0")
    d = deep_narrow_dims()
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=synthetic_state_dict(d, device=dev, dtype=bf), det_token_idx=d.det_token_idx,
                             num_frames=8, pe_dtype=torch.float32)
    batch = synthetic_batch(d, B=1, T=8, L=128, n_det=3, seed=11)
    kw = batch.as_kwargs(inference=True)
    kd = dict(kw)
    for k in ("global_enc_images", "grounding_enc_images"):
        kd[k] = kw[k].to(dev).to(bf)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kd[k] = kw[k].to(dev)
    out = model(**kd)
    sd = LazyRoundedWeights(d, gen_device=dev)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    gi, si = kw["global_enc_images"].to(bf).float(), kw["grounding_enc_images"].to(bf).float()
    with torch.no_grad():
        emb_o = O.sam_image_encoder(sd, d, si)
        feats_o, _ = O.encode_images(sd, d, gi)
        embeds, _, _ = O.splice(sd, kw["input_ids"], None, None, feats_o)
        hidden_o = O.llama_forward(sd, d, embeds, None)
        mask = O.det_token_mask(d, kw["input_ids"])
        pemb_o = O.pred_embeddings(sd, d, hidden_o, mask)
        pe = O.dense_pe(sd, d)
        _, _, box_o, obj_o = O.decode_boxes(sd, d, pemb_o, emb_o, kw["original_size_list"], pe, True)
        # B: oracle decoder on the HIP towers' outputs
        g = d.sam_grid
        emb_h = out["image_embeddings"].float().cpu().view(8, g, g, -1).permute(0, 3, 1, 2).contiguous()
        pemb_h = O.pred_embeddings(sd, d, out["hidden"].float().cpu(), mask)
        _, _, box_b, _ = O.decode_boxes(sd, d, pemb_h, emb_h, kw["original_size_list"], pe, True)
        _, _, box_b1, _ = O.decode_boxes(sd, d, pemb_h, emb_o, kw["original_size_list"], pe, True)
        _, _, box_b2, _ = O.decode_boxes(sd, d, pemb_o, emb_h, kw["original_size_list"], pe, True)
    # A: HIP decoder on the oracle's tower outputs
    emb_rows = emb_o.permute(0, 2, 3, 1).reshape(8 * g * g, -1).to(bf).to(dev).contiguous()
    n_det = 3
    text = torch.cat([pemb_o[t] for t in range(len(pemb_o))], 0).to(bf).to(dev).contiguous()  # (frame, det) order
    inst_frame = torch.arange(8, dtype=torch.int32, device=dev).repeat_interleave(n_det)
    box_a, obj_a, _ = model.decoder.forward(emb_rows, Var(text), inst_frame)
    l1 = lambda a, b: (a.float().cpu() - b).abs().mean().item()  # noqa: E731
    res = {"C_end_to_end": l1(out["flat_boxes"], box_o), "A_hip_decoder_on_oracle_towers": l1(box_a, box_o),
           "B_oracle_decoder_on_hip_towers": l1(box_b, box_o), "B1_only_hip_text": l1(box_b1, box_o), "B2_only_hip_sam": l1(box_b2, box_o)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
