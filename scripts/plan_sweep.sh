# bash scripts/plan_sweep.sh -> loop ms per step under convolution-plan knobs (liso_amd/csrc/conv_mfma.hip make_plan)
export PYTHONPATH=$PWD
run() { env "$@" python bench.py --no-cpu-baseline --no-iou3d --no-legs --steps 48 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"; }
for kv in "$@"; do run $kv; done
