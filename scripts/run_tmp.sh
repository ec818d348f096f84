mkdir -p gpurun_out/r6l
python -m pytest tests/test_gpu_slim.py tests/test_gpu_liso_loop.py tests/test_gpu_flow_io.py tests/test_gpu_parity_full_size.py -q -m gpu > gpurun_out/r6l/tests.txt 2>&1
python scripts/stage_alone_times.py > gpurun_out/r6l/stages.txt 2>&1
python bench.py --no-legs --no-cpu-baseline --no-iou3d --no-fp32-leg > gpurun_out/r6l/bench.txt 2>&1
LISO_UPDATE_MERGED=0 python bench.py --no-legs --no-cpu-baseline --no-iou3d --no-fp32-leg > gpurun_out/r6l/bench_unmerged.txt 2>&1
