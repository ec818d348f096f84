import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=400, use_graph=True, overlap=False)
pairs = [slim_pair(2 + 100 * i, dev) for i in range(2)]
which = sys.argv[1] if len(sys.argv) > 1 else "A"
for i in range(4):
    tr.step(*pairs[i % 2])
torch.cuda.synchronize()
g = tr._infer_graph if which == "A" else tr.detector._graph
for _ in range(40):
    g.replay()
torch.cuda.synchronize()
