"""Round 4: does keeping the CLIP tower in bf16 (`fp8_policy="det16_kv16_clip16"`) keep the fp8 configuration's boxes where the bf16
path's are when the LLaMA stream carries massive-activation channels? Deep-narrow fp8 parity for both policies, without and with
outliers (tools/fp8_policy_study.py --outliers predicted: det16_kv16 1.9e-2 -> 9e-4 with CLIP in bf16).
    python tools/fp8_clip_policy.py  -> gpurun_out/fp8_clip_policy.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_full_depth_gpu as F  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for outl in (0.0, 1000.0):
    for pol in ("det16_kv16", "det16_kv16_clip16"):
        r = F.run_fp8_parity(dev, "deep_narrow", pol, outliers=outl)
        out[f"{pol}@outliers{outl:g}"] = {k: r[k] for k in ("box_l1_vs_oracle", "box_l1_max", "objectness_logit_abs_err", "llama_hidden_rel_rms", "projected_features_rel_rms")}
        torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp8_clip_policy.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
