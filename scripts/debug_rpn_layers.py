"""layer-by-layer forward of the RPN at the train-step fixture's step-1 weights: own fused chain vs fp64 torch chain"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
from keyed_init import keyed_state_dict
from liso_amd.utils.config import default_cfg
from liso_amd.trainer import DetectorTrainer
from liso_amd.utils import mfma_conv as MC

fx = np.load("tests/golden/train_step_reference.npz", allow_pickle=True)
dev = torch.device("cuda:0")
cfg = default_cfg(grid=64, bev_range_m=40.0)
cfg.optimization.num_training_steps = 8
tr = DetectorTrainer(cfg, dev, compute_dtype=torch.float32)
sd = tr.net.state_dict()
init = keyed_state_dict({k: (tuple(v.shape), v.dtype) for k, v in sd.items()})
tr.net.load_state_dict({**sd, **{k: v.to(dev) for k, v in init.items()}}, strict=True)
pcls = [torch.from_numpy(fx["pcl_0"]).to(dev), torch.from_numpy(fx["pcl_1"]).to(dev)]
targets = {k: torch.from_numpy(fx["gt_" + k]).to(dev) for k in ("probs", "rot", "dims", "pos")}
targets["center_bool_mask"] = torch.from_numpy(fx["center_mask"]).to(dev)
rpn = tr.net.model.rpn
grabbed = {}
rpn.register_forward_pre_hook(lambda m, a: grabbed.__setitem__("x", a[0].detach().clone()))
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["LISO_CONV_BACKEND"] = "miopen"
for _ in range(nsteps):
    tr.model.train(); tr.optimizer.zero_grad(set_to_none=True)
    total, _, _ = tr.loss(pcls, targets); total.backward(); tr.optimizer.step(); tr.lr_scheduler.step()
tr.model.train()
with torch.no_grad():
    tr.loss(pcls, targets)
x0 = grabbed["x"]
print("rpn input", tuple(x0.shape), x0.dtype, x0.stride())
os.environ["LISO_CONV_BACKEND"] = "mfma"


def ref_layer(x, conv, bn, spec):
    w = conv.weight.double()
    if spec.transposed:
        y = F.conv_transpose2d(x, w, None, stride=spec.stride)
    else:
        y = F.conv2d(x, w, None, stride=spec.stride, padding=spec.padding)
    m = y.mean((0, 2, 3), keepdim=True); v = y.var((0, 2, 3), unbiased=False, keepdim=True)
    return y, F.relu((y - m) / torch.sqrt(v + bn.eps) * bn.weight.double().view(1, -1, 1, 1) + bn.bias.double().view(1, -1, 1, 1))


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


with torch.no_grad():
    xr = x0.double(); xo, fold = x0, None
    for i, block in enumerate(rpn.blocks):
        mods = list(block)
        layers = [(mods[1], mods[2], MC.ConvSpec(3, 3, mods[1].stride[0], 1))] + [(mods[j], mods[j + 1], MC.ConvSpec.of(mods[j]) if hasattr(MC.ConvSpec, "of") else None) for j in range(4, len(mods), 3)]
        for li, (conv, bn, spec) in enumerate(layers):
            rm = bn.running_mean.clone(); rv = bn.running_var.clone(); nb = bn.num_batches_tracked.clone()
            if spec is None:
                spec = MC.ConvSpec(conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], conv.padding[0])
            yr, xr_next = ref_layer(xr, conv, bn, spec)
            xo_raw, fold_next = MC.fused_conv(xo, fold, conv, out_bn=bn, spec=spec)
            # isolate this layer: feed the reference activation too
            one_raw, one_fold = MC.fused_conv(xr.float().contiguous(memory_format=torch.channels_last), None, conv, out_bn=bn, spec=spec)
            bn.running_mean.copy_(rm); bn.running_var.copy_(rv); bn.num_batches_tracked.copy_(nb)
            print(f"block {i} layer {li} {tuple(yr.shape)} chain raw {rel(xo_raw, yr):.2e} norm {rel(MC.materialize(xo_raw, fold_next), xr_next):.2e} | isolated raw {rel(one_raw, yr):.2e} norm {rel(MC.materialize(one_raw, one_fold), xr_next):.2e}"
                  f" | running_mean |.| {float(rm.abs().max()):.3e}")
            xr, xo, fold = xr_next, xo_raw, fold_next
        d = rpn.deblocks[i]
        conv, bn = d[0], d[1]
        tp = isinstance(conv, torch.nn.ConvTranspose2d)
        spec = MC.ConvSpec(conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], 0, transposed=tp)
        rm = bn.running_mean.clone(); rv = bn.running_var.clone(); nb = bn.num_batches_tracked.clone()
        yr, ur = ref_layer(xr, conv, bn, spec)
        uo_raw, ufold = MC.fused_conv(xo, fold, conv, out_bn=bn)
        bn.running_mean.copy_(rm); bn.running_var.copy_(rv); bn.num_batches_tracked.copy_(nb)
        print(f"deblock {i} {tuple(yr.shape)} chain raw {rel(uo_raw, yr):.2e} norm {rel(MC.materialize(uo_raw, ufold), ur):.2e}")
