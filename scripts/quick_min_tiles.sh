for rep in 1 2 3; do for mt in 4 2; do export LISO_WGRAD_RS3_MIN_TILES=$mt; timeout 600 python bench.py --workload slim --graph --no-cpu-baseline --no-iou3d --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slim rs3 min_tiles $mt', round(d['ms_per_step'],3), d['step_times']['median_ms'], d['roofline']['timed_kernels_ms_per_step'].get('conv_f32x3_wgrad'))"; done; done
