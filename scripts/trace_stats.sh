#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics of one bench command, the rows matching a regex printed.
#   bash scripts/trace_stats.sh <tag> <regex> [bench args...]    -> gpurun_out/<tag>_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; RE=$2; shift; shift
OUT=$R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --no-iou3d --no-legs --steps ${TRACE_STEPS:-50} --warmup 5 "$@" > $OUT/line.json 2> $OUT/err.txt
F=$(find $OUT -name "*_kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/${TAG}_stats.csv
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python3 - "$F" "$RE" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if re.search(sys.argv[2], r["Name"]):
        print(r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), round(float(r["AverageNs"]) / 1e3, 2), r["Name"][:100])
PY
