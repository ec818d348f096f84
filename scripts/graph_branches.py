"""captures the SLIM (or detector) train-step hipGraph with debug mode and reports its shape: root nodes, nodes with several
children (forks), nodes with several parents (joins) -- a capture that stayed on one stream is a single chain"""
import os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair, detector_batch
from liso_amd.trainer import SlimTrainer, DetectorTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

which = sys.argv[1] if len(sys.argv) > 1 else "slim"
dev = torch.device("cuda")
orig = torch.cuda.CUDAGraph


class Dbg(orig):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        g.enable_debug_mode()
        return g


torch.cuda.CUDAGraph = Dbg
if which == "slim":
    cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
    tr = SlimTrainer(cfg, dev, use_graph=True)
    s0, s1 = slim_pair(2, dev)
    tr.capture(s0, s1)
else:
    cfg = default_cfg(grid=512, bev_range_m=100.0)
    tr = DetectorTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=20, use_graph=True)
    pcls, targets = detector_batch(seed=1, batch=4, device=dev)
    tr.step(pcls, targets)
torch.cuda.synchronize()
path = "/tmp/graph.dot"
tr._graph.debug_dump(path)
txt = open(path).read()
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
nodes = set(re.findall(r'^\s*"?(\w+)"?\s*\[', txt, flags=re.M)) | {a for a, _ in edges} | {b for _, b in edges}
children, parents = collections.Counter(a for a, _ in edges), collections.Counter(b for _, b in edges)
roots = [n for n in nodes if parents[n] == 0]
forks = [n for n in nodes if children[n] > 1]
joins = [n for n in nodes if parents[n] > 1]
print(f"{which}: {len(nodes)} nodes, {len(edges)} edges, {len(roots)} roots, {len(forks)} forks, {len(joins)} joins, dot {len(txt)} bytes")
labels = dict(re.findall(r'^\s*"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt, flags=re.M))
for n in (roots[:6] + forks[:12]):
    print("  ", "root" if n in roots else "fork", n, labels.get(n, "?")[:150].replace("\n", " "))
