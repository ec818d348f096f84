run() { # label, env..., args
  label=$1; shift
  env "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'ms', round(d['ms_per_step'],3), 'median', round(d['step_times']['median_ms'],3), 'p90', round(d['step_times']['p90_ms'],3))"
}
B="timeout 200 python bench.py --no-legs --no-fp32-leg --no-cpu-baseline --no-iou3d --steps 40"
for rep in 1 2; do
run "la7_w10" $B --warmup 10 2>/dev/null
run "la9_w14" $B --warmup 14 --lookahead 9 2>/dev/null
run "la9_cus160" LISO_INFER_CUS=160 $B --warmup 14 --lookahead 9 2>/dev/null
run "la11_w18" $B --warmup 18 --lookahead 11 2>/dev/null
run "mine2" LISO_MINE_STREAMS=2 $B --warmup 10 2>/dev/null
run "la5_w10" $B --warmup 10 --lookahead 5 2>/dev/null
done
