"""kernel-trace neighbours of the runtime's memset kernels (__amd_rocclr_fillBuffer*) in a rocprofv3 --kernel-trace CSV"""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if "fillBuffer" in n:
        prev = names[i - 1][:70] if i else "-"
        nxt = names[i + 1][:70] if i + 1 < len(names) else "-"
        ctx[(prev, nxt)] += 1
print("memset kernels:", sum(ctx.values()), "of", len(names))
for (p, n), c in ctx.most_common(25):
    print(f"{c:5d} x  after [{p}]  before [{n}]")
