python -m pytest tests -q -m gpu > gpurun_out/r05_final_tests.txt 2>&1
KRE="conv_roles_kernel|conv_igemm_kernel|conv_wgrad|pfn_|corr_lookup|knn_query|dbscan"
bash scripts/profile_round.sh r05_loop "$KRE" > /dev/null 2>&1
bash scripts/profile_round.sh r05_detector "$KRE" --workload detector > /dev/null 2>&1
bash scripts/profile_round.sh r05_slim "$KRE" --workload slim --graph > /dev/null 2>&1
bash scripts/profile_round.sh r05_stress "$KRE" --workload stress > /dev/null 2>&1
python scripts/slim_infer_layers.py --ib 4 > gpurun_out/r05_slim_infer_layers.txt 2>&1
python scripts/stage_b_launches.py > gpurun_out/r05_stage_b_launches.txt 2>&1
python scripts/stage_alone_times.py > gpurun_out/r05_stage_alone_times.txt 2>&1
