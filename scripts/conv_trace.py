"""per-launch durations of the own conv kernels from a rocprofv3 --kernel-trace csv, grouped by (kernel, grid, lds)"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("conv_igemm", "conv_wgrad", "wgrad_reduce", "pack_weights", "conv_bn_finalize", "bn_bwd")):
        continue
    key = (n.split("(")[0][-60:], r["Grid_Size_X"], r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), r.get("VGPR_Count", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print("%9.1f us avg x%-4d total %8.1f us  grid %-8s lds %-7s vgpr %-4s %s" % (sum(v) / len(v), len(v), sum(v), k[1], k[2], k[3], k[0]))
print("total", tot)
