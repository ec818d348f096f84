"""Does query locality matter?  Same kNN query with the clouds in random order vs sorted by BEV pillar."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.slim.slim_loss.knn_graph import KnnIndex

dev = torch.device("cuda:0")
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
ref = s1["pcl_ta"]["pcl"][0, :, :3].contiguous()
q0 = s0["pcl_ta"]["pcl"][0, :, :3].contiguous()
g = torch.Generator(device="cpu").manual_seed(0)
def timeit(idx, q):
    idx.query(q)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        i, d = idx.query(q, return_dist_sqr=True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3, float(d.sum())
idx = KnnIndex(ref, extent=[-50.0, -50.0, 50.0, 50.0])
for sigma in (0.0, 0.3):
    q = (q0 + torch.randn(120000, 3, generator=g).to(dev) * sigma * torch.tensor([1, 1, 0.2], device=dev)).contiguous()
    cell = ((q0[:, 0] + 50) / 0.2).long() * 512 + ((q0[:, 1] + 50) / 0.2).long()
    order = torch.argsort(cell)
    print("sigma %.1f random order %.1f us | sorted by cell %.1f us" % (sigma, timeit(idx, q)[0], timeit(idx, q[order].contiguous())[0]))
