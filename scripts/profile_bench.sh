#!/bin/bash
# Runs on the GPU box (via gpurun): warm MIOpen's find-db with a plain run, then kernel-trace + two PMC passes.
# Usage: bash scripts/profile_bench.sh <tag> <kernel-regex for the PMC passes> [bench args...]   -> gpurun_out/<tag>/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
KRE=${2:-corr_lookup}
shift; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 "$@" > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_write.err
# keep the summaries only: per-dispatch traces are tens of MB and gpurun_out/ is capped at 64 MiB
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT | tail -1
python3 $R/scripts/kstats.py $OUT/trace "" 2>/dev/null | head -${TOPN:-25}
tail -c 1500 $OUT/bench_plain.json
