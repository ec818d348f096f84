"""aten-op census of ONE HeadDecoder forward and of its backward on the GPU (TorchDispatchMode)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
VIEW = {"view", "slice", "select", "permute", "unsqueeze", "squeeze", "expand", "alias", "detach", "t", "transpose", "as_strided", "_unsafe_view", "reshape", "unbind", "split", "split_with_sizes", "_reshape_alias", "lift_fresh", "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "unsafe_split", "unsafe_chunk"}
class Count(TorchDispatchMode):
    def __init__(self): super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = func.overloadpacket.__name__
        if n not in VIEW: self.c[n] += 1
        return func(*args, **(kwargs or {}))
dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
tr.step(s0, s1)
import liso_amd.slim.model.head_decoder as hd
import liso_amd.slim.slim_loss.slim_loss_adaptor as la
counts = {"decoder": collections.Counter(), "loss": collections.Counter()}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        with Count() as c:
            r = f(*a, **k)
        counts[key].update(c.c); counts[key]["#calls"] += 1
        return r
    setattr(obj, name, g)
wrap(hd.HeadDecoder, "forward", "decoder")
wrap(la, "selfsupervisedSlimSingleScaleLoss", "loss")
tr.model.train()
total, _, _ = tr.loss(s0, s1)
with Count() as cb:
    total.backward()
for k, c in counts.items():
    n = c.pop("#calls")
    print(k, "calls", n, "ops per call %.1f" % (sum(c.values()) / n), [(a, round(b / n, 1)) for a, b in c.most_common(18)])
print("backward total ops", sum(cb.c.values()), cb.c.most_common(25))
