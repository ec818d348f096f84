mkdir -p gpurun_out/r6k
for la in 5 7 9 11; do
  python bench.py --no-legs --no-cpu-baseline --no-iou3d --no-fp32-leg --lookahead $la --steps 60 --warmup 20 > gpurun_out/r6k/la$la.txt 2>&1
done
