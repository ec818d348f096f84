"""kNN query micro-benchmark on the SLIM bench clouds: sweep the z-bin count / cell size of the fine grid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
import liso_amd.slim.slim_loss.knn_graph as kg

dev = torch.device("cuda:0")
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
ref = s1["pcl_ta"]["pcl"][0, :, :3].contiguous()
g = torch.Generator(device="cpu").manual_seed(0)
for sigma in (0.05, 0.5, 2.0):
    qry = (s0["pcl_ta"]["pcl"][0, :, :3] + torch.randn(120000, 3, generator=g).to(dev) * sigma * torch.tensor([1, 1, 0.2], device=dev)).contiguous()
    for cell, nz, zc in ((0.2, 1, 8.0), (0.2, 4, 2.0), (0.2, 8, 1.0), (0.2, 16, 0.5), (0.2, 32, 0.25), (0.3, 8, 1.0), (0.4, 8, 1.0), (0.4, 16, 0.5), (0.4, 1, 8.0)):
        class K(kg.KnnIndex):
            def __init__(self, ref):
                self.ref = ref
                lo, hi = [-50.0, -50.0], [50.0, 50.0]
                self.fine = kg._Grid(ref, lo, hi, cell, 700, -4.0, zc, nz)
                self.coarse = kg._Grid(ref, lo, hi, 2.0, 700, -4.0, 2.0, 4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            idx = K(ref)
        torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 5
        idx.query(qry)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            i, d = idx.query(qry, return_dist_sqr=True)
        b.record(); torch.cuda.synchronize()
        print(f"sigma {sigma:4.2f} cell {cell} nz {nz:2d}: query {a.elapsed_time(b)/10*1e3:7.1f} us  build {tb*1e6:7.1f} us  checksum {float(d.sum()):.4f}")
