"""kernel stats (rocprofv3 --stats csv) grouped into families: python scripts/kfam.py <kernel_stats.csv> [steps]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = {}
def family(n):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    m = re.match(r"(conv_igemm_kernel<\d)", n)
    if m: return m.group(1) + ",...>"
    m = re.match(r"(conv_wgrad_kernel<\d)", n)
    if m: return m.group(1) + ",...>"
    if n.startswith("at::native") or "rocprim" in n or "at_cuda" in n or "Cijk" in n or "cub" in n: return "ATen/rocPRIM/rocBLAS"
    if "miopen" in n.lower() or "naive_conv" in n or "Im2d" in n or "gfx9" in n or "igemm" in n.lower() or "MIOpen" in n: return "MIOpen"
    if n.startswith("__amd_rocclr"): return "runtime copy/fill"
    return n.split("(")[0].split("<")[0]
tot = 0.0
for r in rows:
    f = family(r["Name"])
    a = fam.setdefault(f, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); tot += float(r["TotalDurationNs"])
print(f"total kernel time {tot/1e6:.2f} ms; per step {tot/1e6/steps:.3f} ms")
for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{f[:60]:60s} calls/step {c/steps:7.1f}  us/call {t/1e3/c:7.1f}  ms/step {t/1e6/steps:6.3f}  {100*t/tot:5.1f}%")
