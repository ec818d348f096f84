import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_detector import _setup, _rel
from oracle.train_step import detector_forward_loss
from oracle import pillars as OP
tr, pcls, targets = _setup(128, 100.0, 2, 20000)
sd0 = {k: v.detach().cpu().clone() for k, v in tr.net.state_dict().items()}
tr.model.train()
# hook bev grad
bev_holder = {}
orig = tr.net.model.pfn.forward
def wrapped(pcl_t0, img_t0=None):
    x, occ = orig(pcl_t0, img_t0); x.retain_grad(); bev_holder['x'] = x; return x, occ
tr.net.model.pfn.forward = wrapped
total, losses, _ = tr.loss(pcls, targets); total.backward()
sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k and k != "pillar_center_coors_m" else v.clone()) for k, v in sd0.items()}
ref_total, ref_raw, ref_bev = detector_forward_loss(sd, [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}, 128, 100.0)
ref_bev.retain_grad(); ref_total.backward()
print('loss', float(total), float(ref_total))
print('bev fwd rel', _rel(bev_holder['x'], ref_bev))
print('bev grad rel', _rel(bev_holder['x'].grad, ref_bev.grad))
g1 = bev_holder['x'].grad.cpu(); g2 = ref_bev.grad
occ = (ref_bev.abs().sum(1, keepdim=True) > 0)
print('bev grad rel on occupied', (g1-g2).mul(occ).abs().max().item() / g2.mul(occ).abs().max().item(), 'max grad', g2.abs().max().item())
names = dict(tr.net.named_parameters())
worst = sorted(((_rel(names[k].grad, sd[k].grad), k) for k in names if names[k].grad is not None), reverse=True)[:8]
for w in worst: print(w)
# isolate: feed CPU bev grad into GPU pillar backward
tr.net.zero_grad()
x, occ2 = orig(pcls); (x * ref_bev.grad.cuda()).sum().backward()
pre = "model.pfn.pts_voxel_encoder.pfn_layers.0."
for k in ("linear.weight", "norm.weight", "norm.bias"):
    print('pfn bwd with CPU canvas grad', k, _rel(names[pre + k].grad, sd[pre + k].grad))
