"""Golden vectors for the reference's SAM position-table resizing (train.py:503-558: resize_abs_pos_embedding,
resize_rel_pos_embedding).

Runs the reference's OWN functions (extracted from its source file at generation time — train.py imports deepspeed / peft at
module level, which this container lacks) on seeded inputs and stores inputs and outputs only. Container-only:
/root/reference does not travel.
    python oracle/refgen/make_posembed_golden.py
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/train.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "posembed_resize_seed5.npz")


def main():
    src = open(REF).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ("resize_abs_pos_embedding", "resize_rel_pos_embedding")]
    ns = {"F": F, "torch": torch}
    exec(compile(ast.Module(body=fns, type_ignores=[]), REF, "exec"), ns)
    g = torch.Generator().manual_seed(5)
    out = {}
    # (source grid, target image size, patch, channels): the real case is 64 -> 32 at patch 16; a small one and an up-scaling one beside it
    for tag, (gs, target, patch, c, hd) in {"sam": (64, 512, 16, 4, 8), "small": (8, 64, 16, 16, 8), "up": (4, 128, 16, 8, 16)}.items():
        pe = torch.randn(1, gs, gs, c, generator=g)
        rh, rw = torch.randn(2 * gs - 1, hd, generator=g), torch.randn(2 * gs - 1, hd, generator=g)
        out[tag + "_cfg"] = np.array([gs, target, patch, c, hd])
        out[tag + "_pos_in"] = pe.numpy()
        out[tag + "_pos_out"] = ns["resize_abs_pos_embedding"](pe, target, patch, c).numpy()
        out[tag + "_relh_in"], out[tag + "_relw_in"] = rh.numpy(), rw.numpy()
        oh, ow = ns["resize_rel_pos_embedding"](rh, rw, target, patch, hd)
        out[tag + "_relh_out"], out[tag + "_relw_out"] = oh.numpy(), ow.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
