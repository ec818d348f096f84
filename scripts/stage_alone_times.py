"""GPU time of stage A (SLIM inference graph + eager pillars) and stage C (detector step) of the LISO loop when each runs alone"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=400, use_graph=True, overlap=False)
pairs = [slim_pair(2 + 100 * i, dev) for i in range(2)]
for i in range(6):
    tr.step(*pairs[i % 2])
torch.cuda.synchronize()
N = 50


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(N):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / N


with torch.no_grad():
    flow = tr._infer_flow(*pairs[0]).clone()
    targets, boxes = tr._targets_from_flow(pairs[0][0], flow)
print(f"A alone (pillars eager + inference graph): {timed(lambda i: tr._infer_flow(*pairs[i % 2])):.2f} ms")
print(f"B alone (clustering + NMS + targets, eager): {timed(lambda i: tr._targets_from_flow(pairs[0][0], flow)):.2f} ms")
print(f"C alone (detector step): {timed(lambda i: tr.detector.step(pairs[i % 2][0]['pcl_full_no_ground_ta'], targets)):.2f} ms")
g = tr._infer_graph
print(f"A graph replay only: {timed(lambda i: g.replay()):.2f} ms")
gd = tr.detector._graph
print(f"C graph replay only: {timed(lambda i: gd.replay()):.2f} ms")
