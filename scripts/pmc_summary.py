"""Condenses the rocprofv3 PMC passes of scripts/profile_round.sh into one table per kernel name: launches, mean counter values,
HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB: MI355X_MICROARCH.md) and MFMA busy share.  Also writes
<dir>/pmc_<PASS>.csv with (Kernel_Name, Counter_Name, mean Counter_Value, launches) so the committed file stays small."""
import collections, csv, glob, os, re, sys
out = sys.argv[1]
table = collections.defaultdict(dict)
for pas in ("FETCH_SIZE", "WRITE_SIZE", "MFMA", "LDS"):
    files = glob.glob(os.path.join(out, "pmc_" + pas, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        a = agg[(r["Kernel_Name"], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(out, f"pmc_{pas}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])
        for (k, c), (s, n) in sorted(agg.items()):
            w.writerow([k, c, s / n, n])
            table[k][c] = s / n
            table[k]["launches"] = n
for k, v in sorted(table.items(), key=lambda kv: -kv[1].get("launches", 0)):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:70]
    hbm = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (256 CUs x 4)
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("GRBM_GUI_ACTIVE", 1) / 8.0 * 1024.0, 1)
    line = (f"{name:72s} x{v.get('launches', 0):5d}  HBM {hbm / 1e6:8.2f} MB/launch  MfmaUtil {busy:6.3f}  "
            f"MOPS_BF16 {v.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0):.3g}")
    # LDS column only when BOTH counters were collected (SQ_LDS_IDX_ACTIVE = all LDS-array cycles, MI355X_MICROARCH.md LDS section;
    # round 2 asked for a counter name this part does not have and divided by 1)
    act = v.get("SQ_LDS_IDX_ACTIVE")
    if act and "SQ_LDS_BANK_CONFLICT" in v:
        line += f"  LDS conflict share {v['SQ_LDS_BANK_CONFLICT'] / act:.3f} (conflict cycles / LDS-array cycles)"
    print(line)
