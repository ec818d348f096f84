"""Why the AP sweep of liso_amd/eval/od_metrics.py stays on the host: the recall quotients tp / n_gt computed on the device (float64)
against the host's, for every list of the od_metrics fixture"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.eval import od_metrics as M

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/od_metrics_reference.npz"))
keys = [k[:-7] for k in g.files if k.endswith("_labels") and k[0] in "bi"]
for key in keys:
    lab, sc, fn = (torch.as_tensor(g[key + s]) for s in ("_labels", "_scores", "_is_fn"))
    order = torch.from_numpy(np.argsort(-g[key + "_scores"]))
    keep = order[~fn[order]]
    if keep.numel() == 0:
        continue
    rec = torch.cumsum(lab[keep].double(), 0) / float(lab.sum())
    rec_dev = (torch.cumsum(lab[keep].cuda().double(), 0) / float(lab.sum())).cpu()
    print(key, "recall entries that differ host vs device:", int((rec != rec_dev).sum()), "of", rec.numel(),
          "max |diff|", float((rec - rec_dev).abs().max()))
