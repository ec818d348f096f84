import torch
from liso_amd.datasets.synthetic import detector_batch
from liso_amd.trainer import DetectorTrainer
from liso_amd.utils.config import default_cfg
dev = torch.device("cuda")
torch.manual_seed(0)
tr = DetectorTrainer(default_cfg(grid=256, bev_range_m=50.0), dev, compute_dtype=torch.bfloat16, total_steps=20)
pcls, targets = detector_batch(5, 1, dev, n_points=40000, grid=256, bev_range_m=50.0)
net = tr.net.model
net.train()
with torch.no_grad():
    bev, occ = net.pfn(pcl_t0=pcls, img_t0=None)
bev = bev.detach().clone()
params = [p for p in list(net.rpn.parameters()) + list(net.center_head.parameters())]

def stage(which):
    for p in params:
        p.grad = None
    if which == "rpn":
        raw, fold = net.rpn(bev, lazy=True)
        loss = (raw.float() ** 2).mean()
    elif which == "rpn_mat":
        out = net.rpn(bev)
        loss = (out.float() ** 2).mean()
    else:
        pred = net.center_head(net.rpn(bev, lazy=True))
        loss = sum((v.float() ** 2).mean() for v in pred.values())
    loss.backward()
    gs = [p.grad.clone() if p.grad is not None else None for p in params]
    return loss.detach(), gs

for which in ("rpn", "rpn_mat", "full"):
    ref = stage(which)
    ref = (ref[0].clone(), ref[1])
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        stage(which)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = stage(which)
    g.replay(); torch.cuda.synchronize()
    bad = []
    names = [n for n, _ in list(net.rpn.named_parameters()) + list(net.center_head.named_parameters())]
    for n, a, b in zip(names, out[1], ref[1]):
        if a is None or b is None:
            continue
        e = float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-20))
        if not (e < 1e-2):
            bad.append((n, e))
    print(which, "loss", float(out[0]), float(ref[0]), "bad grads:", bad[:8], len(bad))
