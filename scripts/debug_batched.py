import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
dev = torch.device("cuda")
s0, s1 = slim_pair(60, dev, n_points=30000, grid=256, bev_range_m=50.0)
def run(bd, bdec, tag):
    cfg = default_cfg(grid=256, bev_range_m=50.0)
    cfg = apply_slim_simple_knn_training(cfg) if tag == "simple_knn" else cfg
    torch.manual_seed(0)
    tr = SlimTrainer(cfg, dev)
    tr.net.raft_network.batch_directions = bd
    tr.net.batch_decoding = bdec
    tr.model.train()
    total, _, _ = tr.loss(s0, s1)
    total.backward()
    return float(total), {n: p.grad.clone() for n, p in tr.net.named_parameters() if p.grad is not None}
for tag in ("default", "simple_knn"):
    base = run(False, False, tag)
    for bd, bdec in ((True, False), (False, True), (True, True), (False, False)):
        l, g = run(bd, bdec, tag)
        errs = {k: float((g[k] - base[1][k]).abs().max() / base[1][k].abs().max().clamp(min=1e-30)) for k in g}
        gmax = max(float(v.abs().max()) for v in base[1].values())
        errs = {k: v for k, v in errs.items() if float(base[1][k].abs().max()) > 1e-5 * gmax}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(tag, "dirs", bd, "dec", bdec, "loss", l, base[0], "worst", [(k, "%.2e" % v, "%.2e" % float(base[1][k].abs().max())) for k, v in worst])
