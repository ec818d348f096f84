"""Is a LisoLoopTrainer slower when its streams are later entries of PyTorch's stream pools?  (scripts/two_trainers.py: the trainer created
second runs at 6.6 instead of 4.35 ms per step, even after the first one is gone.)  BURN_LOW / BURN_HIGH: streams taken from the normal /
high-priority pool before the trainer is built."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd.trainer import LisoLoopTrainer  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

dev = torch.device("cuda:0")
keep = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("BURN_LOW", "0")))]
keep += [torch.cuda.Stream(device=dev, priority=-1) for _ in range(int(os.environ.get("BURN_HIGH", "0")))]
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
pairs = [slim_pair(2 + 100 * i, dev, n_points=120000 + (i % 5 - 2) * 1500, grid=512, bev_range_m=100.0) for i in range(16)]
batch, n_up = 2, 11
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=256, use_graph=True, overlap=True, infer_batch=4, flow_ahead=2)
ctr = [0]


def run(steps):
    for _ in range(steps):
        i = ctr[0] * batch
        ctr[0] += 1
        tr.step_batch([pairs[(i + k) % 16] for k in range(batch)], upcoming=tuple(pairs[(i + k) % 16] for k in range(batch, batch + n_up)))


run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(20)
torch.cuda.synchronize()
print(f"BURN_LOW={os.environ.get('BURN_LOW', '0')} BURN_HIGH={os.environ.get('BURN_HIGH', '0')}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step "
      f"(flow stream {tr._flow_stream.cuda_stream:#x}, mine stream {tr._mine_stream.cuda_stream:#x})", flush=True)
