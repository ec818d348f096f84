import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.slim.model.raft_code.utils import upflow_n, uplogits_n
from liso_amd.slim.model.raft_mod import change_flow_convention_from_raft2usfl
from liso_amd.slim.model.raft_outputs import raft_network_outputs
for (n_it, B, dirs, h, w) in [(6, 1, 2, 64, 64), (3, 2, 2, 16, 24)]:
    g = torch.Generator().manual_seed(n_it * 100 + h)
    b2, adapter = dirs * B, 0.1953125
    flows = [torch.randn(b2, 2, h, w, generator=g).cuda().requires_grad_(True) for _ in range(n_it)]
    logits = [torch.randn(b2, 4, h, w, generator=g).cuda().requires_grad_(True) for _ in range(n_it)]
    out = raft_network_outputs(flows, logits, dirs=dirs, factor=8, resolution_adapter=adapter)
    ref = []
    for f, lg in zip(flows, logits):
        up = change_flow_convention_from_raft2usfl(upflow_n(f, n=8), resolution_adapter=adapter)
        ref.append(torch.cat([uplogits_n(lg, n=8), up, up], dim=1).permute(0, 2, 3, 1))
    want = torch.cat([ref[it][d * B:(d + 1) * B] for d in range(dirs) for it in range(n_it)], dim=0)
    # fp64 truth
    ref64 = []
    for f, lg in zip(flows, logits):
        up = change_flow_convention_from_raft2usfl(upflow_n(f.double(), n=8), resolution_adapter=adapter)
        ref64.append(torch.cat([uplogits_n(lg.double(), n=8), up, up], dim=1).permute(0, 2, 3, 1))
    w64 = torch.cat([ref64[it][d * B:(d + 1) * B] for d in range(dirs) for it in range(n_it)], dim=0)
    print("fwd: hip-aten %.3e  hip-fp64 %.3e  aten-fp64 %.3e  max %.3f" % (float((out - want).abs().max()), float((out - w64).abs().max()),
                                                                     float((want - w64).abs().max()), float(want.abs().max())))
    wgt = torch.randn(out.shape, generator=g).cuda()
    (out * wgt).sum().backward()
    got = [(f.grad.clone(), lg.grad.clone()) for f, lg in zip(flows, logits)]
    for t in flows + logits:
        t.grad = None
    (want * wgt).sum().backward()
    aten = [(f.grad.clone(), lg.grad.clone()) for f, lg in zip(flows, logits)]
    for t in flows + logits:
        t.grad = None
    (w64 * wgt.double()).sum().backward()
    d1 = max(float((a[i] - b[i]).abs().max()) for a, b in zip(got, aten) for i in (0, 1))
    d2 = max(float((a[0] - f.grad).abs().max()) for a, f in zip(got, flows))
    d3 = max(float((a[0] - f.grad).abs().max()) for a, f in zip(aten, flows))
    print("bwd: hip-aten %.3e  hip-fp64(flow) %.3e  aten-fp64(flow) %.3e  max %.3f" % (d1, d2, d3, max(float(a[0].abs().max()) for a in aten)))
