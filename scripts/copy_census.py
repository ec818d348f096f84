import torch
from liso_amd.utils import mfma_conv as MC
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=20, use_graph=False)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
tr.step(s0, s1)
MC._COPY_LOG = {}
tr.step(s0, s1)
for k, v in sorted(MC._COPY_LOG.items(), key=lambda kv: -kv[1]):
    print(v, k)
