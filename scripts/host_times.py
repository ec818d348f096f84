"""Is the pipelined LISO loop bound by the host or by the GPU?  Host seconds per step spent enqueueing each stage and waiting for
stage B's results (python scripts/host_times.py [lookahead]); a wait share near zero = the host is the bottleneck."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd import trainer as T  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

lookahead = int(sys.argv[1]) if len(sys.argv) > 1 else 7
flow_ahead, batch = 2, 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = T.LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=256, use_graph=True, overlap=True,
                       infer_batch=max(1, lookahead - 1 - flow_ahead), flow_ahead=flow_ahead)
n_up = batch * (2 + flow_ahead) + max(1, lookahead - 1 - flow_ahead) - 1
n_pairs = max(16, n_up + batch + 2)
pairs = [slim_pair(2 + 100 * i, dev, n_points=120000 + (i % 5 - 2) * 1500, grid=512, bev_range_m=100.0) for i in range(n_pairs)]
acc = collections.Counter()


def wrap(obj, name, key):
    inner = getattr(obj, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            acc[key] += time.perf_counter() - t0
    setattr(obj, name, f)


wrap(tr, "_stage_a", "A enqueue")
wrap(tr, "_stage_b", "B enqueue")
wrap(tr.detector, "step", "C enqueue (pillars + replay + AdamW)")
wrap(tr, "_take_mined", "wait for stage B result")
ctr = [0]


def step():
    i = ctr[0] * batch
    ctr[0] += 1
    return tr.step_batch([pairs[(i + k) % n_pairs] for k in range(batch)], upcoming=tuple(pairs[(i + k) % n_pairs] for k in range(batch, batch + n_up)))


for _ in range(12):
    step()
torch.cuda.synchronize()
acc.clear()
N = 48
t0 = time.perf_counter()
for _ in range(N):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"lookahead {lookahead}: {t_all / N * 1e3:.3f} ms per step; host loop {t_host / N * 1e3:.3f} ms per step")
for k, v in acc.most_common():
    print(f"  {k:42s} {v / N * 1e3:7.3f} ms per step")
