"""where does stage B (capacity mode) still synchronise with the host?  torch.cuda.set_sync_debug_mode('warn') + stack traces"""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=40, use_graph=False)
s0, s1 = slim_pair(2, dev)
with torch.no_grad():
    flow = tr.slim.infer_point_flow_t0_t1(s0, s1)
tr._targets_from_flow(s0, flow, capacity=64)
torch.cuda.synchronize()
seen = {}
def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "liso_amd" in f.filename]
        seen[" < ".join(reversed(st[-4:]))] = seen.get(" < ".join(reversed(st[-4:])), 0) + 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr._targets_from_flow(s0, flow, capacity=64)
torch.cuda.set_sync_debug_mode("default")
for k, v in seen.items():
    print(v, "x", k)
print("sync points:", sum(seen.values()))
import time
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    tr._targets_from_flow(s0, flow, capacity=64)
t_host = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
print(f"host {1e3 * t_host:.2f} ms per call, with GPU {1e3 * (time.perf_counter() - t) / 20:.2f} ms")
