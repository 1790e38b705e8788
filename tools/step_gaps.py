"""GPU idle gaps inside one training step, from a rocprofv3 kernel-trace database (rocpd .db): which kernels the device waits
before, and when in the step. Steps are delimited by the optimizer launches (adamw_*).
    python tools/step_gaps.py gpurun_out/prof/xyz_results.db [step_index]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ad = [(s, e) for n, s, e in rows if "adamw" in n]
bursts, cb = [], [ad[0][0], ad[0][1]]
for s, e in ad[1:]:
    if s - cb[1] > 20e6:
        bursts.append(tuple(cb)); cb = [s, e]
    else:
        cb[1] = e
bursts.append(tuple(cb))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0, t1 = bursts[k][1], bursts[k + 1][1]
step = [(n, s, e) for n, s, e in rows if s >= t0 and e <= t1]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-48:]
gaps, busy_until = [], step[0][2]
for i in range(1, len(step)):
    g = step[i][1] - busy_until
    if g > 0:
        gaps.append((g, short(step[i - 1][0]), short(step[i][0]), (step[i][1] - t0) / 1e6))
    busy_until = max(busy_until, step[i][2])
print(f"step {k}: wall {(t1 - t0) / 1e6:.1f} ms, {len(step)} kernels, kernel sum {sum(e - s for _, s, e in step) / 1e6:.1f} ms, "
      f"idle {sum(g[0] for g in gaps) / 1e6:.2f} ms ({sum(1 for g in gaps if g[0] > 50e3)} gaps > 50 us: {sum(g[0] for g in gaps if g[0] > 50e3) / 1e6:.2f} ms)")
for g in sorted(gaps, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 15]:
    print(f"  {g[0] / 1e3:8.1f} us at {g[3]:7.1f} ms   {g[1]} -> {g[2]}")
h = collections.Counter()
for g, _, _, t in gaps:
    h[int(t // 10)] += g
print("  idle ms per 10-ms window:", {b * 10: round(v / 1e6, 2) for b, v in sorted(h.items()) if v > 0.2e6})
