#!/bin/bash
# own MFMA convolutions vs MIOpen (torch's convolution), same trainers, same launch modes -> gpurun_out/r02_compare/*.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_compare
mkdir -p $O
B="timeout 280 python3 $R/bench.py --no-cpu-baseline --no-iou3d"
$B > $O/loop_own_graph_overlap.json 2>/dev/null
$B --miopen-convs > $O/loop_miopen_graph_overlap.json 2>/dev/null
$B --no-overlap > $O/loop_own_graph.json 2>/dev/null
$B --no-overlap --miopen-convs > $O/loop_miopen_graph.json 2>/dev/null
$B --eager > $O/loop_own_eager.json 2>/dev/null
$B --eager --miopen-convs > $O/loop_miopen_eager.json 2>/dev/null
$B --workload detector > $O/detector_own_graph.json 2>/dev/null
$B --workload detector --miopen-convs > $O/detector_miopen_graph.json 2>/dev/null
$B --workload detector --eager > $O/detector_own_eager.json 2>/dev/null
$B --workload detector --eager --miopen-convs > $O/detector_miopen_eager.json 2>/dev/null
$B --workload slim --graph > $O/slim_own_graph.json 2>/dev/null
$B --workload slim --graph --miopen-convs > $O/slim_miopen_graph.json 2>/dev/null
$B --workload slim > $O/slim_own_eager.json 2>/dev/null
$B --workload slim --miopen-convs > $O/slim_miopen_eager.json 2>/dev/null
for f in $O/*.json; do python3 - "$f" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{os.path.basename(sys.argv[1]):36s} {d['ms_per_step']:8.2f} ms/step  {d['value']:8.1f} frames/s")
except Exception as e:
    print(os.path.basename(sys.argv[1]), "FAILED", e)
PY
done
