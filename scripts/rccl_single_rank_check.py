"""one rank, RCCL process group alive (watchdog thread, communicator on this GPU): do the loop's hipGraph captures and replays coexist with
it, and does an all-reduce of the detector's flat gradient buffer between replays work?  (Two ranks cannot share one GPU under RCCL; the
multi-rank logic itself is covered with gloo in tests/test_gpu_multirank.py.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")  # (the test sets its own port)
import torch
import torch.distributed as dist
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(8, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
cfg = apply_slim_simple_knn_training(default_cfg(grid=256, bev_range_m=50.0))
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=40, use_graph=True)
pairs = [slim_pair(5 + i % 2, dev, n_points=30000, grid=256, bev_range_m=50.0) for i in range(8)]
losses = []
for i in range(10):
    cur = [pairs[(2 * i + k) % 8] for k in range(2)]
    up = [pairs[(2 * i + k) % 8] for k in range(2, 6)]
    losses.append(float(tr.step_batch(cur, upcoming=up)))
    dist.all_reduce(tr.detector._flat_grad)  # what a second rank would add after every replay
torch.cuda.synchronize()
# launch + completion cost of the gradient all-reduce on this 1-rank communicator (no data leaves the GPU: a lower bound of what every
# step adds before any byte crosses xGMI) -- DESIGN.md 9 uses it in the labelled-unmeasured 8-GPU estimate
g = tr.detector._flat_grad
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    dist.all_reduce(g)
torch.cuda.synchronize()
a.record()
for _ in range(50):
    dist.all_reduce(g)
b.record()
torch.cuda.synchronize()
print(f"all_reduce of the flat gradient buffer ({g.numel() * g.element_size() / 1e6:.1f} MB) on the 1-rank RCCL group: "
      f"{a.elapsed_time(b) / 50 * 1e3:.1f} us per call")
dist.barrier()
print("ok: 10 loop steps with captures + replays next to a live RCCL process group; losses", [round(v, 3) for v in losses[:4]], "...")
dist.destroy_process_group()
