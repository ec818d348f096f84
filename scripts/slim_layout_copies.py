"""which tensors does `mfma_conv.as_nhwc` have to copy in one eager SLIM training step (a layout the kernels cannot read in place)?
(shape, strides, the three innermost frames outside mfma_conv) -> count"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd.trainer import SlimTrainer  # noqa: E402
from liso_amd.utils import mfma_conv as MC  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = SlimTrainer(cfg, dev, use_graph=False)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
tr.step(s0, s1)
MC._COPY_LOG = {}
tr.step(s0, s1)
torch.cuda.synchronize()
tot = 0
for (shape, stride, fr), n in sorted(MC._COPY_LOG.items(), key=lambda kv: -kv[1]):
    mb = 4 * n
    for v in shape:
        mb *= v
    tot += mb
    print(f"{n:3d} x {str(shape):24s} strides {str(stride):28s} {mb / 1e6:7.1f} MB  {' <- '.join(reversed(fr))}")
print(f"total copied: {tot / 1e6:.1f} MB")
