"""Outlier stress of the precision claims. Every parity figure of rounds 1-3 was taken on N(0, 0.02)-like weights, which
have no activation outliers; real LLaMA checkpoints carry "massive activations" — a few hidden channels 1e2-1e3 x the typical magnitude at
every position. `synthetic_state_dict(..., outliers=F)` scales the weights that WRITE three fixed channels of the residual stream
(embed_tokens columns, o_proj of layer 0, down_proj of layers 0-1). This script re-runs the deep-narrow (full depth, quarter width) parity
cases of tests/test_full_depth_gpu.py with such weights — inference (fp32 streams), training mode (bf16 streams: losses + gradient groups
vs torch autograd through the oracle) and the two fp8 policies — and writes the figures next to the no-outlier ones.

    python tools/outlier_stress.py --factor 1000 --out gpurun_out/outlier_stress.json
"""
import argparse
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--factor", type=float, default=1000.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "outlier_stress.json"))
    ap.add_argument("--cases", default="inference,training,fp8_all,fp8_det16_kv16")
    ap.add_argument("--which", default="deep_narrow", choices=["deep_narrow", "full"], help="full = FULL dims (inference / fp8 cases: minutes of CPU oracle each)")
    args = ap.parse_args()
    import torch
    spec = importlib.util.spec_from_file_location("full_depth", os.path.join(ROOT, "tests", "test_full_depth_gpu.py"))
    FD = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FD)
    dev = torch.device("cuda:0")
    W = args.which
    out = {"factor": args.factor, "model": ("deep narrow (LLaMA 32 x 1024, CLIP 24 x 256, SAM 32 x 320)" if W == "deep_narrow" else "FULL dims") + ", B=1, T=8, L=128, n_det=3"}
    for F in (0.0, args.factor):
        key = "outliers" if F else "baseline"
        rec = {}
        cases = args.cases.split(",")
        if "inference" in cases:
            r = FD.run_inference_parity(dev, W, outliers=F)
            rec["inference"] = {k: r[k] for k in ("box_l1_vs_oracle_full", "box_l1_max_full", "objectness_logit_abs_err", "llama_hidden_rel_rms",
                                                  "llama_hidden_rel_max", "llama_stream_abs_max_oracle")}
        if "training" in cases:
            r = FD.run_training_parity(dev, W, outliers=F)
            rec["training"] = {"loss_terms_rel_err": r["loss_terms_rel_err"], "whole_gradient": r["whole_gradient"],
                               "gradient_groups": {g: {"cos": round(v["cos"], 5), "norm_ratio": round(v["norm_ratio"], 4)} for g, v in r["gradient_groups"].items()},
                               "box_l1_train_mode_vs_oracle": r["box_l1_train_mode_vs_oracle"]}
        for pol in ("all", "det16_kv16", "det16_kv16_clip16"):
            if "fp8_" + pol in cases:
                r = FD.run_fp8_parity(dev, W, pol, outliers=F)
                rec["fp8_" + pol] = {k: r[k] for k in ("box_l1_vs_oracle", "box_l1_max", "objectness_logit_abs_err", "llama_hidden_rel_rms")}
        out[key] = rec
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
