"""Quick SLIM train-step timing (development helper; bench.py is the contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev)
s0, s1 = slim_pair(1, dev, n_points=120000, grid=512, bev_range_m=100.0)
for i in range(3):
    l = tr.step(s0, s1)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(K):
    l = tr.step(s0, s1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print("SLIM step ms", dt * 1e3, "loss", float(l), "frames/s", 2 / dt, "mem GB", torch.cuda.max_memory_allocated() / 1e9)
