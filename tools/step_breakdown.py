"""Per-kernel time inside ONE training step from a rocprofv3 kernel-trace CSV (steps are delimited by the AdamW launches).
    python tools/step_breakdown.py gpurun_out/prof/x_kernel_trace.csv [step_from_end=2] [rows=45]"""
import collections, csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda x: x[1])
ad = [(s, e) for n, s, e in rows if "adamw" in n]
b, cb = [], list(ad[0])
for s, e in ad[1:]:
    if s - cb[1] > 20e6:
        b.append(tuple(cb)); cb = [s, e]
    else:
        cb[1] = e
b.append(tuple(cb))
k = len(b) - 1 - (int(sys.argv[2]) if len(sys.argv) > 2 else 2)
t0, t1 = b[k][1], b[k + 1][1]
st = [(n, s, e) for n, s, e in rows if s >= t0 and e <= t1]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n).split("(")[0][:84]
c = collections.defaultdict(lambda: [0, 0])
for n, s, e in st:
    c[short(n)][0] += e - s; c[short(n)][1] += 1
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, kernel sum {sum(e - s for _, s, e in st) / 1e6:.2f} ms, {len(st)} launches")
grp = collections.Counter()
for n, (t, _) in c.items():
    g = ("gemm" if n.startswith("gemm") else "attention" if ("attn" in n or "flash" in n or "rel_bias" in n) else "norm" if n.startswith("norm") else
         "torch" if ("elementwise" in n or "Functor" in n or "copyBuffer" in n or "CatArray" in n or "index_" in n or "reduce" in n) else "other")
    grp[g] += t
print("  groups (ms):", {g: round(t / 1e6, 2) for g, t in grp.most_common()})
for n, (t, k2) in sorted(c.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{t / 1e6:8.2f} ms {k2:5d} x {t / k2 / 1e3:8.1f} us  {n}")
