import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from liso_amd.utils.config import default_cfg
from liso_amd.datasets.synthetic import detector_batch
from liso_amd.trainer import DetectorTrainer
import liso_amd.trainer as T
VIEW = {"view", "slice", "select", "permute", "unsqueeze", "squeeze", "expand", "alias", "detach", "t", "transpose", "as_strided", "_unsafe_view", "reshape", "unbind", "split", "split_with_sizes", "_reshape_alias", "lift_fresh", "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided"}
class Count(TorchDispatchMode):
    def __init__(self): super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = func.overloadpacket.__name__
        if n not in VIEW: self.c[n] += 1
        return func(*args, **(kwargs or {}))
dev = torch.device("cuda:0")
cfg = default_cfg(grid=512, bev_range_m=100.0)
tr = DetectorTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=20)
pcls, targets = detector_batch(1, 4, dev, n_points=120000, grid=512, bev_range_m=100.0)
tr.step(pcls, targets)
net = tr.net
orig_fwd = net.model.forward
with Count() as c_all:
    total, losses, _ = tr.loss(pcls, targets)
with Count() as c_net:
    feats = net.model(pcls) if False else None
# split: network part only
with Count() as c_model:
    out = net.model(None, pcls, None) if False else None
with Count() as c_b:
    total.backward()
with Count() as c_o:
    tr.optimizer.step()
print("forward total ops", sum(c_all.c.values()), c_all.c.most_common(30))
print("backward ops", sum(c_b.c.values()), c_b.c.most_common(25))
print("optimizer ops", sum(c_o.c.values()))
