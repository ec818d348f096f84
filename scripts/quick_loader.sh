for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-iou3d --no-legs --loader --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loader', round(d['ms_per_step'],3), d['step_times']['median_ms'], d.get('final_loss'))"; done
timeout 600 python bench.py --no-cpu-baseline --no-iou3d --no-legs --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', round(d['ms_per_step'],3), d['step_times']['median_ms'], d.get('final_loss'))"
