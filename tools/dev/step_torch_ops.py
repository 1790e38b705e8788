"""Which torch-side launches (copies, fills, cats, elementwise) sit inside ONE training step, how long each takes and what runs either side of it.
    python tools/dev/step_torch_ops.py <kernel_trace.csv> [step_from_end=2]"""
import csv, re, sys
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
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
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n).split("(")[0][:60]
tot = 0
for i, (n, s, e) in enumerate(st):
    if any(t in n for t in ("copyBuffer", "Functor", "CatArray", "fillBuffer", "elementwise_kernel", "index_", "reduce_kernel")):
        tot += e - s
        print(f"{(s - t0) / 1e6:8.2f} ms  {(e - s) / 1e3:8.1f} us  {short(n):40s} | after {short(st[i - 1][0]) if i else '-'} | before {short(st[i + 1][0]) if i + 1 < len(st) else '-'}")
print(f"total {tot / 1e6:.2f} ms of a {(t1 - t0) / 1e6:.1f} ms step")
