"""reads a rocprofv3 kernel-trace CSV: GPU busy time (union of kernel intervals), sum of kernel durations, per-queue/stream
shares, over the last `frac` of the trace -- how much do the pipeline's streams really overlap?"""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"]) for r in rows))
t_lo = ks[0][0] + int((ks[-1][1] - ks[0][0]) * float(sys.argv[2]) if len(sys.argv) > 2 else 0)
ks = [k for k in ks if k[0] >= t_lo]
span = ks[-1][1] - ks[0][0]
busy, cur_s, cur_e = 0, ks[0][0], ks[0][1]
for s, e, *_ in ks[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in ks)
print(f"kernels {len(ks)} span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms ({busy/span:.1%})  sum of durations {tot/1e6:.2f} ms  overlap factor {tot/busy:.2f}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, q, st, n in ks:
    by[(q, st)][0] += e - s; by[(q, st)][1] += 1
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  queue {k[0]} stream {k[1]}: {v[0]/1e6:.2f} ms in {v[1]} kernels")
