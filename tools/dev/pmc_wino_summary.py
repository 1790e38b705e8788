"""FETCH_SIZE / WRITE_SIZE per launch of the Winograd pipeline's kernels against their algorithmic bytes.
usage: python tools/dev/pmc_wino_summary.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> > profiles/r06_pmc_winograd_traffic.json"""
import csv, glob, json, os, sys
C, rows, tiles = 1280, 32768, 4096
ALG = {"wino3d_tokens_kernel<0>": (rows * C * 2, 64 * tiles * C * 2), "wino3d_tokens_kernel<1>": (rows * C * 2, 64 * tiles * C * 2),
       "wino3d_weight_kernel": (27 * C * C * 2, 64 * C * C * 2), "gemm_nt_pp_kernel<256, false, -1, false, true>": (64 * tiles * C * 2 + 64 * C * C * 2, 64 * tiles * C * 2),
       "wino3d_output_kernel": (64 * tiles * C * 2 + rows * C * 2, 2 * rows * C * 2), "gemm_tn_pp_kernel<false, false, true>": (2 * 64 * tiles * C * 2, 64 * C * C * 4),
       "wino3d_wgrad_kernel": (64 * C * C * 4 + 27 * C * C * 4, 27 * C * C * 4)}


def per_kernel(root, counter):
    f = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        n = n[:n.index("(")] if "(" in n else n
        if n in ALG:
            out.setdefault(n, []).append(float(r["Counter_Value"]) * 1024)
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/dev/pmc_wino.py; gfx950 tallies 128-byte read requests at 64 B: fetch x 2 "
               "(MI355X_MICROARCH.md, HBM section); bytes per launch, mean of 3", "kernels": {}}
for k, (ar, aw) in ALG.items():
    if k in fetch and k in write:
        fb, wb = 2 * fetch[k], write[k]
        res["kernels"][k] = {"fetch_MB": round(fb / 1e6, 1), "write_MB": round(wb / 1e6, 1), "algorithmic_read_MB": round(ar / 1e6, 1), "algorithmic_write_MB": round(aw / 1e6, 1),
                             "traffic_over_algorithmic": round((fb + wb) / (ar + aw), 2)}
print(json.dumps(res, indent=1))
