"""lists the ATen ops of one SLIM training step (plan passed in) that are backed by rocPRIM / hipCUB device algorithms"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

SUS = ("sort", "cumsum", "cumprod", "nonzero", "masked_select", "index_put", "unique", "kthvalue", "topk", "searchsorted", "bucketize",
       "histc", "bincount", "median", "mode", "index_add", "scatter_reduce", "masked_scatter", "randperm", "multinomial", "logcumsumexp",
       "cummax", "cummin", "argsort", "msort", "embedding", "index.Tensor", "repeat_interleave", "roll", "take", "put")


class Scan(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in SUS):
            n = max([a.numel() for a in args if torch.is_tensor(a)] + [0])
            self.c[(name, n)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = SlimTrainer(cfg, dev, use_graph=True)
tr.model.train()
s0, s1 = slim_pair(2, dev)
canv = tuple(c.detach().clone() for c in tr._pillars(s0, s1))
with torch.no_grad():
    tr.net(s0, s1, None, canvases=canv)
plan = tr.net.build_gather_plan(s0, s1, *tr.net.gather_plan_meta)
with Scan() as sc:
    total, _, _ = tr.loss(s0, s1, (True, True), canvases=canv, gather_plan=plan)
    fwd = dict(sc.c); sc.c.clear()
    total.backward()
print("forward + loss:")
for k, v in sorted(fwd.items(), key=lambda kv: -kv[0][1]):
    print("  ", v, "x", k[0], "max input numel", k[1])
print("backward:")
for k, v in sorted(sc.c.items(), key=lambda kv: -kv[0][1]):
    print("  ", v, "x", k[0], "max input numel", k[1])
