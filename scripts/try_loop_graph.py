import sys, time, torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
grid, rng, n = (int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 50.0, 40000)
dev = torch.device("cuda")
pairs = [slim_pair(7 + i, dev, n_points=n, grid=grid, bev_range_m=rng) for i in range(2)]
out = []
import os
MODE = os.environ.get('GMODE', 'both')
for use_graph in (False, {'both': True, 'infer': 'infer', 'detector': 'detector'}[MODE]):
    cfg = apply_slim_simple_knn_training(default_cfg(grid=grid, bev_range_m=rng))
    torch.manual_seed(0)
    tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=20, use_graph=use_graph)
    losses = []
    for i in range(6):
        losses.append(float(tr.step(*pairs[i % 2])))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        tr.step(*pairs[i % 2])
    torch.cuda.synchronize()
    print("graph" if use_graph else "eager", [round(l, 4) for l in losses], "ms/step", round((time.perf_counter() - t0) * 100, 2), "boxes", int(tr.last_boxes.valid.sum()), flush=True)
    out.append(losses)
print("max rel diff", max(abs(a - b) / abs(a) for a, b in zip(*out)))
