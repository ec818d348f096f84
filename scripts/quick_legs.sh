for w in slim detector; do timeout 600 python bench.py --workload $w --graph --no-cpu-baseline --no-iou3d --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],3), d['step_times']['median_ms'], d['roofline']['timed_kernels_ms_per_step'].get('bn_bwd'))"; done
timeout 600 python bench.py --no-cpu-baseline --no-iou3d --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loop', round(d['ms_per_step'],3), d['step_times']['median_ms'])"
