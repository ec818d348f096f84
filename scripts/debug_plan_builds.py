import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
from liso_amd.slim.slim_loss import static_aggregation as SA

dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = SlimTrainer(cfg, dev, use_graph=True)
tr.model.train()
s0, s1 = slim_pair(2, dev)
orig_sort = torch.sort
def sort(*a, **k):
    n = a[0].numel()
    if n > 100000:
        fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[-6:-1]]
        print("torch.sort of", n, "from", " < ".join(reversed(fr)), flush=True)
    return orig_sort(*a, **k)
torch.sort = sort
canv = tuple(c.detach().clone() for c in tr._pillars(s0, s1))
with torch.no_grad():
    tr.net(s0, s1, None, canvases=canv)
plan = tr.net.build_gather_plan(s0, s1, *tr.net.gather_plan_meta)
print("---- with a plan passed in", flush=True)
total, _, _ = tr.loss(s0, s1, (True, True), canvases=canv, gather_plan=plan)
print("---- backward", flush=True)
total.backward()
