KRE="conv_roles_kernel|conv_igemm_kernel|conv_wgrad|pfn_|corr_lookup|knn_query|dbscan"
bash scripts/profile_round.sh r05_loop "$KRE" > /dev/null 2>&1
bash scripts/profile_round.sh r05_detector "$KRE" --workload detector > /dev/null 2>&1
bash scripts/profile_round.sh r05_slim "$KRE" --workload slim --graph > /dev/null 2>&1
bash scripts/profile_round.sh r05_stress "$KRE" --workload stress > /dev/null 2>&1
du -sh gpurun_out/r05_*
