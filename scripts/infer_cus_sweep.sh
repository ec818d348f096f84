for rep in 1 2; do for c in 0 128 160 96; do
LISO_INFER_CUS=$c timeout 200 python bench.py --no-legs --no-fp32-leg --no-cpu-baseline --no-iou3d --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cus', $c, 'ms', round(d['ms_per_step'],3), 'median', round(d['step_times']['median_ms'],3), 'p90', round(d['step_times']['p90_ms'],3), 'frac', round(d['roofline']['frac'],3), 'avg_launch_ms', round(d['roofline']['avg_launch_ms'],4))"
done; done
