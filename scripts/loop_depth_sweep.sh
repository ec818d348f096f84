B="timeout 200 python bench.py --no-legs --no-fp32-leg --no-cpu-baseline --no-iou3d --steps 40"
for rep in 1 2; do for cfg in "7 128 10 2" "9 128 14 2" "9 96 14 2" "9 112 14 2" "11 96 18 2" "11 112 18 2" "9 128 14 3" "7 128 10 3"; do set -- $cfg
LISO_INFER_CUS=$2 $B --warmup $3 --lookahead $1 --flow-ahead $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lookahead $1 cus $2 flow_ahead $4', 'ms', round(d['ms_per_step'],3), 'median', round(d['step_times']['median_ms'],3), 'captures in timed region', d['config']['graph_captures']['inside_timed_region'])"
done; done
