#!/bin/bash
# bash scripts/pmc_one.sh <tag> <kernel-regex> -- python3 script args... : two SQ counter passes of one command, kernel-filtered means
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; KRE=$2; shift; shift; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-include-regex "$KRE" --output-format csv -d $OUT/p1 -- "$@" > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --kernel-include-regex "$KRE" --output-format csv -d $OUT/p2 -- "$@" > /dev/null 2> $OUT/p2.err
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    fs = glob.glob("$OUT/" + p + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no output"); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(agg.items()):
        print(f"{k:34s} mean {s / n:14.1f}  (n={n})")
PY
find $OUT -name "*.db" -delete
