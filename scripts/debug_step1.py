"""step-1 gradients of the train-step fixture: own convolutions vs torch's on the same weights / running statistics"""
import os, sys, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from keyed_init import keyed_state_dict
from liso_amd.utils.config import default_cfg
from liso_amd.trainer import DetectorTrainer

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
named = dict(tr.net.named_parameters())
keys = [str(k) for k in fx["param_keys"]]


def grads(backend):
    os.environ["LISO_CONV_BACKEND"] = backend
    state = copy.deepcopy(tr.net.state_dict())
    tr.model.train()
    tr.optimizer.zero_grad(set_to_none=True)
    total, _, _ = tr.loss(pcls, targets)
    total.backward()
    g = {k: named[k].grad.clone() for k in keys}
    tr.net.load_state_dict(state)
    return float(total), g


for step in range(2):
    la, ga = grads("miopen")
    lb, gb = grads("mfma")
    ref = fx[f"step{step}_grad_norms"]
    print("step", step, "loss miopen", la, "mfma", lb, "ref", float(fx[f"step{step}_loss"]))
    for i, k in enumerate(keys):
        na, nb = float(ga[k].norm()), float(gb[k].norm())
        d = float((ga[k] - gb[k]).norm()) / max(na, 1e-20)
        if ref[i] > 1e-4 * ref.max():
            print(f"  {k:70s} ref {ref[i]:.4e} miopen {na:.4e} mfma {nb:.4e} |diff|/|g| {d:.2e}")
    os.environ["LISO_CONV_BACKEND"] = "miopen"
    tr.model.train(); tr.optimizer.zero_grad(set_to_none=True)
    total, _, _ = tr.loss(pcls, targets); total.backward(); tr.optimizer.step(); tr.lr_scheduler.step()
