"""Wall-clock split of one fused LISO iteration (synchronised between stages)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # seeds the MIOpen db before torch initialises MIOpen
import torch
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.datasets.targets import render_center_targets
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.nms_iou import perform_nms_on_shapes

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=40)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(4):
    tr.step(s0, s1)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for it in range(6):
    t0 = sync()
    with torch.no_grad():
        preds_fw, _ = tr.slim(s0, s1, None)
        flow = preds_fw[-1].aggregated_flow
    t1 = sync()
    sample = dict(s0); sample[cfg.data.flow_source] = {**s0.get(cfg.data.flow_source, {}), "flow_ta_tb": flow}
    with torch.no_grad():
        boxes = tr.cluster_detector(sample, global_step=1)
    t2 = sync()
    with torch.no_grad():
        if boxes.shape[1] > 0:
            boxes = perform_nms_on_shapes(boxes, max_num_boxes=tr.post_nms, overlap_threshold=tr.nms_iou, pre_nms_max_num_boxes=tr.pre_nms)
            boxes.set_padding_val_to(0.0)
    t3 = sync()
    out = tuple(int(g) // 4 for g in cfg.data.img_grid_size)
    targets = render_center_targets(boxes.pos.float(), boxes.dims.float().clamp(min=1e-3), boxes.rot.float(), boxes.valid, out, tuple(cfg.data.bev_range_m))
    t4 = sync()
    tr.detector.step(s0["pcl_full_no_ground_ta"], targets)
    t5 = sync()
    for k, v in (("slim_fwd", t1 - t0), ("cluster", t2 - t1), ("nms", t3 - t2), ("targets", t4 - t3), ("detector_step", t5 - t4)):
        acc.setdefault(k, []).append(v * 1e3)
for k, v in acc.items():
    print(f"{k:14s} {sorted(v)[len(v)//2]:8.2f} ms")
print("boxes", int(boxes.valid.sum()))
