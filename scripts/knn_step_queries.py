"""the kNN query as the SLIM training step issues it: 6 iterations x 120k warped points in bucket order against one 120k-point cloud
(720k queries per launch) -> us per launch, with the default grid of liso_amd/slim/slim_loss/knn_graph.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.slim.slim_loss.knn_graph import KnnIndex

dev = torch.device("cuda:0")
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
ext = [-50.0, -50.0, 50.0, 50.0]
CELL = float(os.environ.get("KNN_CELL", "0.2"))  # fine cell size (the default of KnnIndex: 0.2 m)
ref = KnnIndex(s1["pcl_ta"]["pcl"][0, :, :3].contiguous(), cell=CELL, extent=ext, all_rows_finite=True)
own = KnnIndex(s0["pcl_ta"]["pcl"][0, :, :3].contiguous(), cell=CELL, extent=ext, all_rows_finite=True)
g = torch.Generator(device="cpu").manual_seed(0)
p0 = s0["pcl_ta"]["pcl"][0, :, :3]
for sigma in (0.05, 0.3, 1.0):
    q = torch.cat([(p0 + torch.randn(120000, 3, generator=g).to(dev) * sigma * torch.tensor([1, 1, 0.2], device=dev))[own.sorted_ids()] for _ in range(6)], 0).contiguous()
    ref.query(q)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        i, d = ref.query(q, return_dist_sqr=True)
    b.record(); torch.cuda.synchronize()
    print(f"flow noise {sigma:4.2f} m: {a.elapsed_time(b) / 10 * 1e3:7.1f} us per 720k queries   checksum {float(d.double().sum()):.6f} {int(i.sum())}")
