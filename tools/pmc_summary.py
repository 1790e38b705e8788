"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean of every counter over the dispatches of each kernel.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [name filter] > summary.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            short = name.replace("void ", "").replace("(anonymous namespace)::", "")
            short = short[:short.index("(")] if "(" in short else short
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["dispatches"] = max(len(v) for v in cs.values())
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in d:
                d[c + "_share_of_wave_cycles"] = round(d[c] / wc, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        d["mfma_busy_over_sq_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"], 4)
    out[k] = d
print(json.dumps(out, indent=1))
