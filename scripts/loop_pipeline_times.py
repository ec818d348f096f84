"""host-side wall time of the pipeline stages of LisoLoopTrainer.step (overlap=True): where does the host spend a step?"""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer, DetectorTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

dev = torch.device("cuda")
overlap = "--no-overlap" not in sys.argv
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
IB = 4 if "--ib4" in sys.argv else 2
FA = 2 if "--fa2" in sys.argv else 0
NB = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1  # sweep pairs per detector step
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=400, use_graph=True, overlap=overlap, infer_batch=IB, flow_ahead=FA)
NP = NB * (2 + FA) + IB - 1 + NB + 2
pairs = [slim_pair(2 + 100 * i, dev) for i in range(NP)]
acc = collections.defaultdict(float)


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[label] += time.perf_counter() - t
        return r
    setattr(obj, name, w)


timed(tr, "_stage_a"); timed(tr, "_stage_b"); timed(tr, "_targets_from_flow"); timed(tr, "_infer_flow"); timed(tr, "_mine_from_graph")
timed(tr, "_take_mined", "_take_mined (host waits for stage B of this pair)")
timed(tr.detector, "step", "detector.step"); timed(tr.detector, "_pillars", "detector._pillars")
timed(tr.detector.optimizer, "step", "optimizer.step")
timed(tr.cluster_detector, "forward", "cluster_detector")
import liso_amd.utils.nms_iou as NI
timed(NI, "perform_nms_on_shapes_padded", "nms")
N = 50
if "--fake-b" in sys.argv:  # upper bound: stage B costs neither host nor GPU time (reuses the first mined result)
    cache = {}
    real = tr._targets_from_flow

    tr._graph_mine = False

    def fake(sample_t0, flow, capacity=None, odom_minus_eye=None):
        if "r" not in cache:
            cache["r"] = real(sample_t0, flow, capacity=capacity)
            tr.cluster_detector.last_num_labels = tr.cluster_detector.last_num_labels.clone()
        return cache["r"]
    tr._targets_from_flow = fake
NUP = NB * (2 + FA) + IB - 1
cur = lambda i: [pairs[(i * NB + k) % NP] for k in range(NB)]
up = lambda i: tuple(pairs[(i * NB + k) % NP] for k in range(NB, NB + NUP))
for i in range(16):
    tr.step_batch(cur(i), upcoming=up(i))
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for i in range(16, 16 + N):
    tr.step_batch(cur(i), upcoming=up(i))
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"overlap={overlap} batch={NB} ms/step {1e3 * t_all / N:.2f} (host loop {1e3 * t_host / N:.2f}) = {1e3 * t_all / N / NB:.2f} ms per pair")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {1e3 * v / N:7.3f} ms/step")
