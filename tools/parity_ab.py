"""Which of the round-4 LLaMA changes (round 6: which form of the SAM Conv3d adapters) moves the deep-narrow training parity figures, and by
how much?  (one subprocess per arm: the switches are read at import)   python tools/parity_ab.py [--outliers 1000] [--arms ...] [--seeds ...]
[--out name] -> gpurun_out/<name>.json"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = {"all_off": dict(GROVE_LLAMA_TAIL="0", GROVE_ROPE_TABLE="0", GROVE_FUSE_ROPE_BWD="0"),
        "tail": dict(GROVE_LLAMA_TAIL="1", GROVE_ROPE_TABLE="0", GROVE_FUSE_ROPE_BWD="0"),
        "table": dict(GROVE_LLAMA_TAIL="0", GROVE_ROPE_TABLE="1", GROVE_FUSE_ROPE_BWD="0"),
        "fuse": dict(GROVE_LLAMA_TAIL="0", GROVE_ROPE_TABLE="0", GROVE_FUSE_ROPE_BWD="1"),
        "all_on": dict(GROVE_LLAMA_TAIL="1", GROVE_ROPE_TABLE="1", GROVE_FUSE_ROPE_BWD="1"),
        # round 6: the SAM Conv3d adapters as 27-tap implicit GEMMs / in Winograd F(2x2x2, 3x3x3) form, everything else at its default
        "conv_direct": dict(GROVE_SAM_WINOGRAD="0"), "conv_wino_wgrad": dict(GROVE_SAM_WINOGRAD="wgrad"), "conv_winograd": dict(GROVE_SAM_WINOGRAD="fwd,dgrad,wgrad")}

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import test_full_depth_gpu as F
r = F.run_training_parity(torch.device("cuda:0"), "deep_narrow", outliers=%f, seed=%d)
print("RESULT " + json.dumps({"box_l1": r["box_l1_train_mode_vs_oracle"], "loss_rel": r["loss_terms_rel_err"], "whole": r["whole_gradient"],
                               "groups": {g: [round(v["cos"], 5), round(v["norm_ratio"], 4)] for g, v in r["gradient_groups"].items()}}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outliers", type=float, nargs="*", default=[0.0, 1000.0])
    ap.add_argument("--arms", nargs="*", default=list(ARMS))
    ap.add_argument("--seeds", type=int, nargs="*", default=[11])
    ap.add_argument("--out", default="parity_ab")
    a = ap.parse_args()
    out = {}
    for o in a.outliers:
      for seed in a.seeds:
        for arm in a.arms:
            env = dict(os.environ, **ARMS[arm])
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT, o, seed)], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            key = f"{arm}@{o:g}/seed{seed}"
            out[key] = json.loads(line[0][7:]) if line else {"error": p.stderr[-800:]}
            print(key, json.dumps(out[key])[:300], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", a.out + ".json"), "w"), indent=1)


if __name__ == "__main__":
    main()
