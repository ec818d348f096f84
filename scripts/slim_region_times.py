"""GPU time (HIP events) of the regions of one eager SLIM train step after the batching."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
import liso_amd.slim.model.head_decoder as hd
import liso_amd.slim.model.raft_mod as rm
import liso_amd.slim.slim_loss.slim_loss_adaptor as la
import liso_amd.trainer as T

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(3):
    tr.step(s0, s1)
ev = {}
def timed(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(); r = f(*a, **k); a1.record()
        ev.setdefault(key, []).append((a0, a1))
        return r
    setattr(obj, name, g)
timed(rm.RAFT, "forward", "raft_forward")
timed(hd.HeadDecoder, "forward", "decoder_forward")
orig = la.selfsupervisedSlimSingleScaleLoss
import liso_amd.slim.slim_loss.slim_loss_adaptor as mod
def lw(*a, **k):
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record(); r = orig(*a, **k); a1.record(); ev.setdefault("loss_forward", []).append((a0, a1)); return r
mod.selfsupervisedSlimSingleScaleLoss = lw
for it in range(3):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tr.model.train()
    e[0].record()
    total, _, _ = tr.loss(s0, s1)
    e[1].record()
    tr.optimizer.zero_grad(set_to_none=True)
    total.backward()
    e[2].record()
    tr.optimizer.step()
    e[3].record()
    torch.cuda.synchronize()
    print("forward %.2f ms | backward %.2f ms | optimizer %.2f ms" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])))
torch.cuda.synchronize()
for k, v in ev.items():
    t = [a.elapsed_time(b) for a, b in v][-len(v) // 3:]
    print("%-18s %.2f ms per step (%d calls)" % (k, sum(t), len(t)))
