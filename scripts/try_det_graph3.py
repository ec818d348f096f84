import os, torch
from liso_amd.datasets.synthetic import detector_batch
from liso_amd.trainer import DetectorTrainer
from liso_amd.utils.config import default_cfg
dev = torch.device("cuda")
exp = os.environ.get("EXP", "base")
batch = detector_batch(5, 1, dev, n_points=40000, grid=256, bev_range_m=50.0)
for use_graph in (False, True):
    torch.manual_seed(0)
    tr = DetectorTrainer(default_cfg(grid=256, bev_range_m=50.0), dev, compute_dtype=torch.bfloat16, total_steps=20, use_graph=use_graph,
                         fused_loss=False if exp == "nofused" else None)
    if exp == "fixedbev":
        with torch.no_grad():
            bev, occ = tr.net.model.pfn(pcl_t0=batch[0], img_t0=None)
        bev, occ = bev.clone(), occ.clone()
        tr.net.model.pfn.forward = lambda pcl_t0, img_t0=None: (bev, occ)
    if exp == "nograph_flat" and use_graph:
        # same flat-grad setup, but run the step eagerly
        tr.model.train(); tr._flat_grad.zero_(); total, _, _ = tr.loss(*batch); total.backward(); print("eager-with-flat", float(total)); continue
    l = [float(tr.step(*batch)) for _ in range(3)]
    print(exp, "graph" if use_graph else "eager", [round(v, 3) for v in l], flush=True)
