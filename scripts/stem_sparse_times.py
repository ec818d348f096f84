"""the encoders' first convolution (7x7 / 2, 64 -> 32 on the 512^2 pillar canvas) on real canvases of the bench's sweeps, with and without
the occupancy map: us per call (hipGraph timing), for the tile heights LISO_CONV_MI selects"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils import mfma_conv as MC
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=10)
from liso_amd.slim.model.slim import get_network_input_pcls
rows, occs = [], []
for i in range(4):
    s0, s1 = slim_pair(3 + i, dev, n_points=120000, grid=512, bev_range_m=100.0)
    with torch.no_grad():
        canv = tr.slim.raft_network.encode_pillars(get_network_input_pcls(cfg, s0, "ta", to_device=dev), get_network_input_pcls(cfg, s1, "ta", to_device=dev))
    rows += [canv[0], canv[2]]; occs += [canv[1], canv[3]]
x = torch.cat(rows, 0).contiguous(memory_format=torch.channels_last)
occ = torch.cat(occs, 0).contiguous()
print("canvas", tuple(x.shape), "occupied cells %.2f %%" % (100 * float((occ > 0).float().mean())))
fnet = tr.slim.raft_network.fnet


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


with torch.no_grad():
    for mi in ("2", "1"):
        os.environ["LISO_CONV_MI"] = mi
        t_dense = timeit(lambda: MC.conv_in(x, None, fnet.conv1, fnet.norm1))
        t_sparse = timeit(lambda: MC.conv_in(x, None, fnet.conv1, fnet.norm1, occupancy=occ))
        print(f"tile rows {4 * int(mi)}: dense {t_dense:7.1f} us   with occupancy {t_sparse:7.1f} us")
