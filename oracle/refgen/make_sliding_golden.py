"""Golden vectors for the sliding-window frame sampler of the reference (infer_iground.py:110-148).

Runs the reference's OWN function (extracted from its source file at generation time and executed here — the file
itself imports cv2/ffmpeg/bleach at module level, which this container lacks) for a range of clip lengths and stores
inputs and outputs only. Container-only: /root/reference does not travel.
    python oracle/refgen/make_sliding_golden.py
"""
import ast
import json
import os

REF = "/root/reference/infer_iground.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "sliding_segments.json")


def main():
    src = open(REF).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "sliding_segment_with_mask")
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    f = ns["sliding_segment_with_mask"]
    cases = []
    for nf in (8, 9, 13, 15, 16, 17, 24, 31, 32, 40, 47, 48, 50, 63, 64, 100):
        for ns_ in (8,):
            idx, masks = f(num_frames=nf, num_segments=ns_)
            cases.append({"num_frames": nf, "num_segments": ns_, "all_indices": idx, "masks": masks})
    json.dump({"source": "infer_iground.py:110-148 sliding_segment_with_mask", "cases": cases}, open(OUT, "w"))
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
