"""Memory-side bytes per launch of the pipelined GEMMs from two rocprofv3 --pmc passes over tools/pmc_gemm.py.
usage: python tools/pmc_gemm_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_pmc_gemm_traffic.json
Dispatches are matched to shapes in launch order (tools/pmc_gemm.py: 3 launches per shape); the stream-K fix-up launch that
follows a split launch is charged to it."""
import csv
import glob
import json
import os
import sys

# (M, N, K, launches per step) of tools/pmc_gemm.py's shapes (profiles/*_gemm_shapes.txt)
SHAPES = [(32768, 1280, 5120, 56), (32768, 5120, 1280, 24), (32768, 3840, 1280, 28), (2812, 11008, 4096, 32), (32768, 1280, 1280, 21),
          (2812, 4096, 22016, 32), (2812, 12288, 4096, 32), (2812, 4096, 11008, 32), (2812, 4096, 4096, 64), (18464, 1024, 4096, 23)]


def per_launch(root, counter):
    f = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "gemm_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []  # [kernel, bytes, fixup bytes]
    for r in rows:
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = name[:name.index("(")]
        kb = float(r["Counter_Value"])
        if "fixup" in name:
            out[-1][2] += kb * 1024
        else:
            out.append([name, kb * 1024, 0.0])
    return out


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
assert len(fetch) == len(write) == 3 * len(SHAPES), (len(fetch), len(write))
shapes, wsum = [], {}
for i, (M, N, K, n) in enumerate(SHAPES):
    fs, ws = fetch[3 * i:3 * i + 3], write[3 * i:3 * i + 3]
    kern = fs[0][0]
    fb = 2 * sum(x[1] + x[2] for x in fs) / 3  # gfx950 tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, HBM section)
    wb = sum(x[1] + x[2] for x in ws) / 3
    alg = 2 * (M * K + N * K + M * N)
    shapes.append({"M": M, "N": N, "K": K, "kernel": kern, "launches_per_step": n, "stream_k_fixup": fs[0][2] > 0,
                   "fetch_bytes_corrected": round(fb), "write_bytes": round(wb), "algorithmic_bytes": alg, "ratio": round((fb + wb) / alg, 2)})
    a = wsum.setdefault(kern, [0.0, 0.0, 0])
    a[0] += n * (fb + wb)
    a[1] += n * alg
    a[2] += n
out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--output-format csv) over tools/pmc_gemm.py: 3 launches per "
                 "shape. Both counters are KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM) and "
                 "counts L2-miss (fabric) traffic, Infinity-Cache hits included. A stream-K launch includes its fix-up launch (the fp32 parts "
                 "written by the main kernel and read back). Means are weighted by launches per step.",
       "shapes": shapes,
       "launch_weighted_mean_bytes": {k: round(v[0] / v[2]) for k, v in wsum.items()},
       "launch_weighted_mean_algorithmic_bytes": {k: round(v[1] / v[2]) for k, v in wsum.items()}}
for k in list(out["launch_weighted_mean_bytes"]):
    short = k[k.index("<"):k.index(",")] + ">"
    out["launch_weighted_mean_bytes"][short] = out["launch_weighted_mean_bytes"][k]
print(json.dumps(out, indent=1))
