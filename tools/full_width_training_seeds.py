"""Full-width (LLaMA-7B / CLIP-L / SAM-H dims) training-mode parity over batch seeds: the configuration the headline bench times, forward +
backward against torch autograd through the fp32 oracle (tests/test_full_depth_gpu.py::run_training_parity, ~50 GB of host memory and
~2 minutes of oracle per seed).   python tools/full_width_training_seeds.py 12 13  -> gpurun_out/full_width_training_seeds.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_full_depth_gpu as F  # noqa: E402

seeds = [int(a) for a in sys.argv[1:]] or [12, 13]
out = {}
for sd in seeds:
    r = F.run_training_parity(torch.device("cuda:0"), "full", seed=sd)
    out[f"seed{sd}"] = {"box_l1_train_mode": r["box_l1_train_mode_vs_oracle"], "loss_terms_rel_err": r["loss_terms_rel_err"], "whole_gradient": r["whole_gradient"],
                        "groups": {g: [round(v["cos"], 5), round(v["norm_ratio"], 4)] for g, v in r["gradient_groups"].items()}}
    print(sd, json.dumps(out[f"seed{sd}"])[:400], flush=True)
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_width_training_seeds.json"), "w"), indent=1)
