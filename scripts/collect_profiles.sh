#!/bin/bash
# bash scripts/collect_profiles.sh r06 loop parity detector slim stress : gpurun_out/<round>_<w>/ (scripts/profile_round.sh) -> profiles/<round>_<w>_*
R=$(cd "$(dirname "$0")/.." && pwd)
RND=$1; shift
for w in "$@"; do
  S=$R/gpurun_out/${RND}_$w
  [ -d "$S" ] || { echo "no $S"; continue; }
  cp $S/bench_line.json $R/profiles/${RND}_${w}_bench_line.json
  # (gpurun MERGES into gpurun_out/: traces of earlier calls stay next to the new one -- take the newest)
  cp "$(ls -t $(find $S/trace -name "*kernel_stats.csv") | head -1)" $R/profiles/${RND}_${w}_kernel_stats.csv
  for k in FETCH_SIZE WRITE_SIZE MFMA LDS; do [ -f $S/pmc_$k.csv ] && cp $S/pmc_$k.csv $R/profiles/${RND}_${w}_pmc_$k.csv; done
  cp $S/pmc_summary.txt $R/profiles/${RND}_${w}_pmc_summary.txt
  echo "$w: $(wc -c < $R/profiles/${RND}_${w}_kernel_stats.csv) bytes of kernel stats"
done
