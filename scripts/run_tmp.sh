mkdir -p gpurun_out/r6h
python -m pytest tests/test_gpu_pillars.py tests/test_gpu_edge_cases.py tests/test_gpu_canaries.py tests/test_gpu_detector.py -q -m gpu > gpurun_out/r6h/tests.txt 2>&1
bash scripts/quick_stats.sh r6h_loop --no-legs --steps 50 --warmup 5 > gpurun_out/r6h/q.txt 2>&1
