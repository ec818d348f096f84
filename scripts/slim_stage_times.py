"""Wall-clock split of one SLIM train step (synchronised between stages) + launch counts per stage (torch profiler)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(3):
    tr.step(s0, s1)

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

import liso_amd.slim.slim_loss.slim_loss_adaptor as A
from liso_amd.slim.slim_loss.knn_graph import KnnIndex
acc = {}
for it in range(5):
    tr.model.train()
    t0 = sync()
    preds_fw, preds_bw = tr.model(s0, s1, None)
    t1 = sync()
    pc1, m1 = s0["pcl_ta"]["pcl"], s0["pcl_ta"]["pcl_is_valid"]
    pc2, m2 = s1["pcl_ta"]["pcl"], s1["pcl_ta"]["pcl_is_valid"]
    ext = [float(v) for v in tr.bev_extent]
    idx1 = [KnnIndex(pc1[0][m1[0]][:, :3], extent=ext)]
    idx2 = [KnnIndex(pc2[0][m2[0]][:, :3], extent=ext)]
    total = torch.zeros(1, device=dev)
    for pfw, pbw in zip(preds_fw, preds_bw):
        total = total + A.selfsupervisedSlimSingleScaleLoss(pc1=pc1, valid_mask_pc1=m1, pc2=pc2, valid_mask_pc2=m2, pred_fw=pfw, pred_bw=pbw,
            moving_thresh_module=tr.net.moving_dynamicness_threshold, loss_cfg=tr.slim_cfg.losses.unsupervised, model_cfg=tr.slim_cfg.model,
            bev_extent=tr.bev_extent, metrics_collector={}, knn_index_pc1=idx1, knn_index_pc2=idx2)
    t2 = sync()
    tr.optimizer.zero_grad(set_to_none=True)
    total.backward()
    t3 = sync()
    tr.optimizer.step(); tr.lr_scheduler.step()
    t4 = sync()
    for k, v in (("forward", t1 - t0), ("loss", t2 - t1), ("backward", t3 - t2), ("optimizer", t4 - t3)):
        acc.setdefault(k, []).append(v * 1e3)
for k, v in acc.items():
    print(f"{k:10s} {sorted(v)[len(v)//2]:8.2f} ms")

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(s0, s1)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
