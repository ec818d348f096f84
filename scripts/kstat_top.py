"""python scripts/kstat_top.py <rocprofv3 output dir> [steps] [n] -> the n largest kernels of a --kernel-trace --stats run: share, launches per step, mean duration"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:n]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.2f}% {int(r["Calls"]) / steps:6.1f}/st {float(r["AverageNs"]) / 1e3:7.1f}us {r["Name"][:110]}')
