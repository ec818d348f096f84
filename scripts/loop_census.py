"""ATen op census + wall / GPU time per stage of one fused LISO iteration"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.datasets.targets import render_center_targets
from liso_amd.trainer import LisoLoopTrainer


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.c[str(func).replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=40)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(4):
    tr.step(s0, s1)


def stage(name, fn, census=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    c = None
    if census:
        with Census() as cz:
            fn()
        c = cz.c
    print(f"== {name}: host {1e3*(t1-t0):.2f} ms, until GPU idle {1e3*(t2-t0):.2f} ms" + (f", {sum(c.values())} aten ops" if c else ""))
    if c:
        print("   " + ", ".join(f"{k} x{v}" for k, v in c.most_common(28)))
    return out


with torch.no_grad():
    flow = stage("slim inference", lambda: tr.slim.infer_point_flow_t0_t1(s0, s1))
    sample = dict(s0)
    sample[cfg.data.flow_source] = {**s0.get(cfg.data.flow_source, {}), "flow_ta_tb": flow}
    boxes = stage("flow cluster detector", lambda: tr.cluster_detector(sample, global_step=1))
    from liso_amd.utils.nms_iou import perform_nms_on_shapes_padded
    boxes = stage("nms", lambda: perform_nms_on_shapes_padded(boxes, max_num_boxes=tr.post_nms, overlap_threshold=tr.nms_iou,
                                                               pre_nms_max_num_boxes=tr.pre_nms))
    out = tuple(int(g) // 4 for g in cfg.data.img_grid_size)
    targets = stage("targets", lambda: render_center_targets(boxes.pos.float(), boxes.dims.float().clamp(min=1e-3), boxes.rot.float(),
                                                              boxes.valid, out, tuple(cfg.data.bev_range_m)))
det = tr.detector


def fwd_bwd():
    det.model.train()
    det.optimizer.zero_grad(set_to_none=True)
    total, _, _ = det.loss(s0["pcl_full_no_ground_ta"], targets)
    total.backward()
    return total


stage("detector fwd+bwd", fwd_bwd)
stage("optimizer", lambda: (det.optimizer.step(), det.lr_scheduler.step()))
