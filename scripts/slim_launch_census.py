"""How many kernel launches / how much GPU time each region of the SLIM forward costs (torch profiler, one step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer
import liso_amd.slim.model.raft_mod as rm
import liso_amd.slim.model.head_decoder as hd
import liso_amd.slim.slim_loss.slim_loss_adaptor as la
import liso_amd.slim.model.update as up
import liso_amd.slim.model.extractor as ex

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(3):
    tr.step(s0, s1)

def wrap(cls, name, label):
    f = getattr(cls, name)
    def g(*a, **k):
        with record_function(label):
            return f(*a, **k)
    setattr(cls, name, g)

wrap(ex.SmallEncoder, "forward", "R_encoder")
wrap(up.SmallUpdateBlock, "forward", "R_update_block")
wrap(hd.HeadDecoder, "forward", "R_decoder")
wrap(rm.CorrBlock, "__call__", "R_corr_lookup")
orig = la.selfsupervisedSlimSingleScaleLoss
def loss_w(*a, **k):
    with record_function("R_loss"):
        return orig(*a, **k)
import liso_amd.trainer as T
la.selfsupervisedSlimSingleScaleLoss = loss_w

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.model.train()
    with record_function("R_forward_all"):
        total, _, _ = tr.loss(s0, s1)
    tr.optimizer.zero_grad(set_to_none=True)
    with record_function("R_backward_all"):
        total.backward()
    with record_function("R_optimizer"):
        tr.optimizer.step()
    torch.cuda.synchronize()

ev = prof.events()
regions = [e for e in ev if e.name.startswith("R_")]
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
launches = [e for e in ev if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipModuleLaunchKernel", "hipExtLaunchKernel")]
import collections
cnt = collections.Counter(); 
for r in regions:
    t0, t1 = r.time_range.start, r.time_range.end
    n = sum(1 for l in launches if t0 <= l.time_range.start <= t1)
    cnt[r.name] += n
    cnt[r.name + "#calls"] += 1
    cnt[r.name + "#cpu_us"] += (t1 - t0)
for k in sorted(set(k.split("#")[0] for k in cnt)):
    print(f"{k:20s} calls {cnt[k+'#calls']:4d}  launches {cnt[k]:6d}  cpu {cnt[k+'#cpu_us']/1e3:8.2f} ms")
print("all launches", len(launches), "device kernels", len(kern), "device time %.1f ms" % (sum(e.time_range.end - e.time_range.start for e in kern) / 1e3))
# top backward nodes by launch count
bw = [e for e in ev if e.name.startswith("autograd::engine::evaluate_function")]
c2 = collections.Counter(); t2 = collections.Counter()
for e in bw:
    t0, t1 = e.time_range.start, e.time_range.end
    c2[e.name.split(": ")[-1]] += sum(1 for l in launches if t0 <= l.time_range.start <= t1)
    t2[e.name.split(": ")[-1]] += 1
for k, v in c2.most_common(25):
    print(f"  bwd {k:45s} nodes {t2[k]:5d} launches {v:6d}")
