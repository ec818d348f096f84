"""kNN query sweep under the SLIM bench's conditions: 6 x 120k queries (the other sweep's points + ~0.5 m flow), answered in
the query cloud's bucket order; fine-grid cell size / z bins / ring budget."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
import liso_amd.slim.slim_loss.knn_graph as kg

dev = torch.device("cuda:0")
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
ref = s1["pcl_ta"]["pcl"][0, :, :3].contiguous()
base = s0["pcl_ta"]["pcl"][0, :, :3].contiguous()
perm = kg.KnnIndex(base, extent=[-50, -50, 50, 50], all_rows_finite=True).sorted_ids()
g = torch.Generator(device="cpu").manual_seed(0)
for sigma in (0.1, 0.5, 1.0):
    q = torch.cat([base + torch.randn(120000, 3, generator=g).to(dev) * sigma * torch.tensor([1, 1, 0.2], device=dev) for _ in range(6)])
    qs = torch.cat([q[i * 120000:(i + 1) * 120000][perm] for i in range(6)]).contiguous()
    for cell, nz, zc, rings in ((0.2, 32, 0.25, 6), (0.3, 32, 0.25, 6), (0.4, 32, 0.25, 4), (0.4, 16, 0.5, 4), (0.5, 16, 0.5, 4), (0.6, 16, 0.5, 3),
                                (0.8, 16, 0.5, 3), (0.3, 16, 0.5, 4), (0.4, 32, 0.25, 8)):
        class K(kg.KnnIndex):
            FINE_RINGS = rings
            def __init__(self, ref):
                self.ref = ref
                lo, hi = [-50.0, -50.0], [50.0, 50.0]
                self.fine = kg._Grid(ref, lo, hi, cell, 700, -4.0, zc, nz)
                self.coarse = kg._Grid(ref, lo, hi, 2.0, 700, -4.0, 2.0, 4)
        idx = K(ref)
        idx.query(qs)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            i, d = idx.query(qs, return_dist_sqr=True)
        b.record(); torch.cuda.synchronize()
        print(f"sigma {sigma:4.2f} cell {cell} nz {nz:2d} rings {rings}: query {a.elapsed_time(b)/10*1e3:7.1f} us  checksum {float(d.sum()):.3f}")
