"""Print the per-kernel averages of a rocprofv3 kernel_stats.csv (our kernels first)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "anonymous namespace"
for r in csv.DictReader(open(f)):
    if pat in r["Name"] and (pat == "" or ("at::native" not in r["Name"] and "ck::" not in r["Name"])):
        print("%9.1f us avg  x%-5s %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["Percentage"]), r["Name"][:100]))
