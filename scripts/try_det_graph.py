import os, sys, torch
from liso_amd.datasets.synthetic import detector_batch
from liso_amd.trainer import DetectorTrainer
from liso_amd.utils.config import default_cfg
dev = torch.device("cuda")
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
batches = [detector_batch(5 + i, 1, dev, n_points=40000, grid=256, bev_range_m=50.0) for i in range(2)]
res = []
for use_graph in (False, True):
    torch.manual_seed(0)
    tr = DetectorTrainer(default_cfg(grid=256, bev_range_m=50.0), dev, compute_dtype=dt, total_steps=20, use_graph=use_graph)
    losses = [float(tr.step(*batches[i % 2])) for i in range(6)]
    print("graph" if use_graph else "eager", [round(l, 4) for l in losses], flush=True)
    res.append(losses)
