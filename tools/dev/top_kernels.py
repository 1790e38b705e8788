"""Top kernels of a rocprofv3 *kernel_stats.csv by total time. usage: top_kernels.py <csv> [n]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("total ms", round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, 2))
for r in rows[:n]:
    print("%9.2f ms %7d x %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
