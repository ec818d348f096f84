for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-iou3d --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('loop', round(d['ms_per_step'],3), r['frac'], r['avg_launch_ms'], r['launches_per_step'], d['step_roofline']['frac'])"; done
