"""Does an IDLE LisoLoopTrainer slow an active one down in the same process?  (bench.py's parity legs did: 7.0-7.7 vs 5.55 ms per step.)
Trainer 1 alone -> trainer 2 next to the idle trainer 1 -> trainer 2 after trainer 1 is deleted -> trainer 2 next to idle extra streams."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd.trainer import LisoLoopTrainer  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
pairs = [slim_pair(2 + 100 * i, dev, n_points=120000 + (i % 5 - 2) * 1500, grid=512, bev_range_m=100.0) for i in range(16)]
batch, n_up = 2, 11


def make():
    torch.manual_seed(0)
    return LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=256, use_graph=True, overlap=True, infer_batch=4, flow_ahead=2)


def run(tr, steps, ctr):
    for _ in range(steps):
        i = ctr[0] * batch
        ctr[0] += 1
        tr.step_batch([pairs[(i + k) % 16] for k in range(batch)], upcoming=tuple(pairs[(i + k) % 16] for k in range(batch, batch + n_up)))


def timed(tr, ctr, label):
    run(tr, 10, ctr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(tr, 20, ctr)
    torch.cuda.synchronize()
    print(f"{label}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step", flush=True)


t1, c1 = make(), [0]
timed(t1, c1, "trainer 1 alone")
t2, c2 = make(), [0]
timed(t2, c2, "trainer 2 next to the idle trainer 1")
timed(t1, c1, "trainer 1 again (trainer 2 idle)")
del t1
gc.collect()
torch.cuda.empty_cache()
timed(t2, c2, "trainer 2 after trainer 1 was deleted")
extra = [torch.cuda.Stream(device=dev) for _ in range(4)]
for s in extra:
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
timed(t2, c2, "trainer 2 next to 4 idle extra streams")
