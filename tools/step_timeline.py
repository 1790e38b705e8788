"""Kernel timeline of one training step inside a time window, from a rocprofv3 kernel-trace database (rocpd .db): start offset,
duration, queue / stream and name of every kernel — which launches share the device, which run alone.
    python tools/step_timeline.py results.db [step_index] [t_lo_ms] [t_hi_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
rows = db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
ad = [(r[1], r[2]) for r in rows if "adamw" in r[0]]
bursts, cb = [], [ad[0][0], ad[0][1]]
for s, e in ad[1:]:
    if s - cb[1] > 20e6:
        bursts.append(tuple(cb)); cb = [s, e]
    else:
        cb[1] = e
bursts.append(tuple(cb))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
t0, t1 = bursts[k][1], bursts[k + 1][1]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]
print(f"columns: {cols}; step {k}: {(t1 - t0) / 1e6:.2f} ms")
for r in rows:
    s, e = r[1], r[2]
    if s < t0 or e > t1:
        continue
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    if b < lo or a > hi:
        continue
    print(f"{a:9.3f} ms  {(e - s) / 1e3:8.1f} us  q={r[3] if qcol else '-'}  {short(r[0])}")
