#!/bin/bash
# bash scripts/quick_stats.sh <tag> [bench args...] -> gpurun_out/<tag>_kernel_stats.csv: rocprofv3 --kernel-trace --stats of one bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-iou3d --no-fp32-leg "$@" > $OUT/bench_traced.json 2> $OUT/trace.err
find $OUT -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \;
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cut -c1-300 $OUT/bench_traced.json
