"""Generates tests/golden/state_dict_keys.json: the parameter KEY SETS of the reference model (/root/reference, imported in this
container only) with and without the temporal-objectness head (mask_decoder.py:83-87), at the tiny test dims. A consolidated
checkpoint written by grove_amd for an ANet / VidSTG run (use_temp_objectness=False, train.py:203) must carry exactly the second
set: infer_anet.py:556 loads it with strict=True.

    python oracle/refgen/make_keys_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_harness as R  # noqa: E402
from grove_amd.synthetic import TINY  # noqa: E402


def main():
    out = {}
    for tag, use in (("with_objectness", True), ("without_objectness", False)):
        model, _sd = R.build_reference_model(TINY, use_temp_objectness=use)
        keys = sorted(model.state_dict().keys())
        out[tag] = keys
        print(tag, len(keys), "keys;", sum("temporal_objectness_head" in k for k in keys), "objectness keys")
    only = sorted(set(out["with_objectness"]) - set(out["without_objectness"]))
    assert only == ["model.grounding_encoder.mask_decoder.temporal_objectness_head.bias",
                    "model.grounding_encoder.mask_decoder.temporal_objectness_head.weight"], only
    assert not set(out["without_objectness"]) - set(out["with_objectness"])
    path = os.path.join(REPO, "tests", "golden", "state_dict_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
