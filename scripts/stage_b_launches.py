"""Stage B (flow clustering -> boxes -> Kabsch -> NMS -> targets) of one sweep pair, launched eagerly with HIP-event timing of every
C-ABI call (`_lib.TIMER`), plus the sizes that drive its kernels: dynamic pillars, labelled pillars, clusters, surviving boxes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd import _lib as L
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=400, use_graph=False, overlap=False, infer_batch=1)
pair = slim_pair(2, dev)
with torch.no_grad():
    flow = tr._infer_flow(pair[0], pair[1])
    for _ in range(3):
        tr._targets_from_flow(pair[0], flow, capacity=tr.box_capacity)
    torch.cuda.synchronize()
    L.TIMER.enable_all()
    L.TIMER.reset()
    for _ in range(5):
        targets, boxes = tr._targets_from_flow(pair[0], flow, capacity=tr.box_capacity)
    torch.cuda.synchronize()
    L.TIMER.disable_all()
cd = tr.cluster_detector
tot = 0.0
for k in L.TIMER.events:
    d = L.TIMER.durations_ms(k)
    if d:
        print(f"{k:28s} {len(d) // 5:3d} launches per pair, {1e3 * sum(d) / 5:8.1f} us per pair")
        tot += 1e3 * sum(d) / 5
print(f"sum of timed C-ABI calls: {tot:.1f} us per pair; clusters {int(cd.last_num_labels.max())}, boxes {int(boxes.valid.sum())}")
for name in ("last_bev_labels",):
    t = getattr(cd, name, None)
    if t is not None:
        print(name, "nonzero cells:", int((t != 0).sum()))
