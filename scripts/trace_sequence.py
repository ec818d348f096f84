"""rocprofv3 kernel trace -> the kernel SEQUENCE of one step on the busiest queue (name, duration, gap to the previous kernel): the
dependent chain of a captured step as the GPU ran it.  usage: trace_sequence.py <dir> <step index from the end, default 3> <kernels per step>"""
import csv
import glob
import re
import sys
import collections

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
by = collections.defaultdict(list)
for r in rows:
    by[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
q = max(by, key=lambda k: sum(e - s for s, e, _ in by[k]))
ks = sorted(by[q])
# a step starts at the kernel named by argv[3] (default: the optimizer launch ends a step)
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_flat_kernel"
ends = [i for i, k in enumerate(ks) if marker in k[2]]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
seq = ks[lo:hi]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:70]


tot, gaps = 0, 0
print(f"queue {q}: {len(seq)} kernels between two {marker} launches, span {(seq[-1][1] - seq[0][0]) / 1e3:.1f} us")
agg = collections.OrderedDict()
prev_end = seq[0][0]
for s, e, n in seq:
    tot += e - s
    gaps += max(0, s - prev_end)
    a = agg.setdefault(short(n), [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += max(0, s - prev_end)
    prev_end = e
print(f"sum of kernel durations {tot / 1e3:.1f} us, sum of gaps {gaps / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {v[0]:3d} x  kernel {v[1] / 1e3:7.1f} us  + gaps in front {v[2] / 1e3:6.1f} us   {k}")
