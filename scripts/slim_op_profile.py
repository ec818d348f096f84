"""torch.profiler table of one eager SLIM train step: device time per aten op / autograd node (where the glue launches come from)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401  (seeds the MIOpen user db before torch initialises)
import torch
from torch.profiler import ProfilerActivity, profile
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(4):
    tr.step(s0, s1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=False) as prof:
    for _ in range(3):
        tr.step(s0, s1)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get("ROWS", 45)), max_name_column_width=60))
