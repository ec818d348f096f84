"""Micro-benchmark of the pillar path only (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import detector_batch
from liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid import PointsPillarFeatureNetWrapper
from liso_amd.utils.config import default_cfg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pcls, _ = detector_batch(1, B, dev, n_points=120000, grid=512, bev_range_m=100.0)
m = PointsPillarFeatureNetWrapper(default_cfg()).to(dev)
m.out_dtype = torch.bfloat16
m.train()
g = None
for it in range(25):
    bev, occ = m(pcls)
    if g is None:
        g = torch.randn_like(bev)
    bev.backward(g)
torch.cuda.synchronize()
print("done", bev.shape)
