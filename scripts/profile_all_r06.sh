#!/bin/bash
# round-6 evidence: bench line + rocprofv3 kernel stats + 4 PMC passes per workload -> gpurun_out/r06_<workload>/ (copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-/root/repo}
KRE='conv_|corr_|pfn_|dbscan|bn_|in_fwd|in_bwd|in_stats|knn_|kabsch|region|bev_|wgrad|stem_|rows_|gru_|raft_|voxel|assign|cell_|seg_|fit_z|nms|pair_matrix|targets|centerloss|adamw|rmsprop|gather_f32|multi_copy|residual|channel_ext'
bash $R/scripts/profile_round.sh r06_loop "$KRE"
bash $R/scripts/profile_round.sh r06_parity "$KRE" --dtype f32x3 --warmup 10
bash $R/scripts/profile_round.sh r06_detector "$KRE" --workload detector
bash $R/scripts/profile_round.sh r06_slim "$KRE" --workload slim --graph
bash $R/scripts/profile_round.sh r06_stress "$KRE" --workload stress
