"""Summarise a rocprofv3 --pmc run per kernel: mean of every counter over the dispatches of each kernel. Reads the
*_counter_collection.csv files (--output-format csv) or, when rocprofv3 wrote its default rocpd database (*_results.db), the
`pmc_events` view of that (a counter's instances — XCDs / SEs — are summed per dispatch first, as the CSV form does).
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv or *_results.db> [name filter] > summary.json"""
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
for f in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):
    import sqlite3
    con = sqlite3.connect(f)
    per = defaultdict(float)
    for name, disp, ctr, val in con.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        if flt and flt not in name:
            continue
        per[(name, disp, ctr)] += float(val)
    for (name, disp, ctr), val in per.items():
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short[:short.index("(")] if "(" in short else short
        acc[short][ctr].append(val)
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
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_of_lds_cycles"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    out[k] = d
print(json.dumps(out, indent=1))
