#!/bin/bash
# Runs on the GPU box (via gpurun): the round's evidence for one bench command.
#   bash scripts/profile_round.sh <tag> <kernel regex for the PMC passes> [bench args...]   -> gpurun_out/<tag>/
# plain bench line | rocprofv3 --kernel-trace --stats | separate PMC passes: FETCH_SIZE, WRITE_SIZE, MFMA counters, LDS.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
KRE=${2:-conv_igemm_kernel}
shift; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py "$@" > $OUT/bench_line.json 2> $OUT/bench.err
# (100 timed steps: the statistics are then those of the steady state -- the synthetic sweeps are generated on the GPU with framework
# kernels, which at 20 steps are a fifth of all framework launches of the process)
rm -rf $OUT/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-iou3d --no-fp32-leg --no-legs --steps ${TRACE_STEPS:-100} --warmup 5 "$@" > $OUT/bench_traced.json 2> $OUT/trace.err
P="--steps 3 --warmup 2 --no-cpu-baseline --no-iou3d --no-fp32-leg --no-legs"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc_FETCH_SIZE -- python3 $R/bench.py $P "$@" > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc_WRITE_SIZE -- python3 $R/bench.py $P "$@" > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc_MFMA -- python3 $R/bench.py $P "$@" > /dev/null 2> $OUT/pmc_mfma.err
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc_LDS -- python3 $R/bench.py $P "$@" > /dev/null 2> $OUT/pmc_lds.err
# keep the summaries only: per-dispatch traces are tens of MB and gpurun_out/ is capped at 64 MiB
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
# (the per-dispatch counter files are summarised into pmc_<PASS>.csv above: drop them, gpurun_out/ is capped at 64 MiB for the WHOLE call)
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT | tail -1
tail -c 600 $OUT/bench_line.json
