"""per-kernel GPU time per step from a rocprofv3 kernel-trace CSV, restricted to the last `frac` of the trace (the timed
steps), grouped by a shortened kernel name.  usage: trace_steps.py <dir> <steps in window> [frac]"""
import csv, sys, glob, collections, re
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ks[0][0], ks[-1][1]
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
ks = [k for k in ks if t0 + (t1 - t0) * lo <= k[0] <= t0 + (t1 - t0) * hi]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    m = re.match(r"at::native::(\w+)<.*?(\w+Functor|\w+_kernel\w*|\w+Op\w*)", n)
    if n.startswith("at::native::"):
        return "aten:" + n[12:90]
    return n[:90]


agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ks:
    a = agg[short(n)]
    a[0] += e - s; a[1] += 1
tot = sum(a[0] for a in agg.values())
print(f"window {(ks[-1][1]-ks[0][0])/1e6:.1f} ms, {len(ks)} kernels, GPU time {tot/1e6:.2f} ms = {tot/1e6/steps:.3f} ms/step over {steps} steps")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[5]) if len(sys.argv) > 5 else 45]:
    print(f"{t/1e3/steps:9.1f} us/step  x{c/steps:6.1f}  avg {t/1e3/c:7.1f} us  {n}")
