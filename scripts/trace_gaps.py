"""rocprofv3 kernel trace -> idle time of every queue inside the last `frac` of the trace: where does the detector's queue (the one with
the most kernel time) wait, and for how long per step?  usage: trace_gaps.py <dir> <steps in window> [frac=0.5] [min_gap_us=8]"""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = float(sys.argv[2])
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
min_gap = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 8e3
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows)
t_lo = ks[0][0] + int((ks[-1][1] - ks[0][0]) * (1 - frac))
ks = [k for k in ks if k[0] >= t_lo]
span = ks[-1][1] - ks[0][0]
by = collections.defaultdict(list)
for s, e, q, n in ks:
    by[q].append((s, e, n))
print(f"window {span / 1e6:.2f} ms = {span / 1e6 / steps:.3f} ms per step over {steps:.0f} steps")
for q, lst in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in lst)
    gaps = [(lst[i + 1][0] - lst[i][1], lst[i][2], lst[i + 1][2]) for i in range(len(lst) - 1) if lst[i + 1][0] - lst[i][1] > 0]
    big = [g for g in gaps if g[0] >= min_gap]
    small = sum(g[0] for g in gaps if g[0] < min_gap)
    print(f"queue {q}: {len(lst) / steps:.0f} kernels / step, busy {busy / 1e6 / steps:.3f} ms / step, gaps < {min_gap / 1e3:.0f} us: {small / 1e6 / steps:.3f} ms / step, "
          f"gaps >= that: {sum(g[0] for g in big) / 1e6 / steps:.3f} ms / step in {len(big) / steps:.1f} gaps / step")
    agg = collections.Counter()
    for g, before, after in big:
        agg[(before[:60], after[:60])] += g
    for (b, a), t in agg.most_common(6):
        print(f"      {t / 1e6 / steps:.3f} ms / step   after [{b}]  before [{a}]")
