#!/bin/bash
# bash scripts/wgrad_sweep.sh -> per-kernel times (rocprofv3 kernel trace) of the weight gradient of one layer under plan knobs
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R
OUT=$R/gpurun_out/wgrad_sweep; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env..., then layer args
  tag=$1; shift
  rm -rf $OUT/t_$tag
  env "$@" true 2>/dev/null
  ( export "${ENVV[@]}"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- python3 $R/scripts/one_conv.py $LAYER wgrad > /dev/null 2> $OUT/err_$tag.txt )
  echo "== $tag ${ENVV[*]} layer $LAYER"
  python3 $R/scripts/kstats.py $OUT/t_$tag wgrad | head -4
}
for LAYER in "64 64 256 3 1 1" "128 128 128 3 1 1" "256 256 64 3 1 1"; do
  ENVV=(X=1); run base
  ENVV=(LISO_WGRAD_BLOCKS=128); run b128
  ENVV=(LISO_WGRAD_TH=4); run th4
  ENVV=(LISO_WGRAD_RS3=0); run old
done
