# static CU partitions of the pipelined loop: inference cap I / detector cap D (persistent 3x3 convolutions only); 0 = uncapped
B="timeout 200 python bench.py --no-legs --no-fp32-leg --no-cpu-baseline --no-iou3d --steps 40 --warmup 10"
for rep in 1 2; do for pair in "128 0" "128 128" "96 160" "112 144" "128 192" "96 192" "0 0"; do set -- $pair
LISO_INFER_CUS=$1 LISO_DETECTOR_CUS=$2 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer $1 detector $2', 'ms', round(d['ms_per_step'],3), 'median', round(d['step_times']['median_ms'],3), 'p90', round(d['step_times']['p90_ms'],3))"
done; done
