"""which source lines issue the ATen launches of a stage?  (op, innermost liso_amd frame) -> count, output MB"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer

VIEW = {"view", "permute", "detach", "slice", "select", "unsqueeze", "expand", "squeeze", "transpose", "t", "as_strided", "alias",
        "_unsafe_view", "reshape", "unbind", "split", "empty", "empty_like", "empty_strided", "lift_fresh", "split_with_sizes", "unfold",
        "_local_scalar_dense", "is_same_size", "sym_size", "sym_stride", "sym_numel", "new_empty", "new_empty_strided"}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.defaultdict(lambda: [0, 0.0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "").split(".")[0]
        if name not in VIEW:
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "liso_amd" in fr.filename and "site-packages" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            mb = out.numel() * out.element_size() / 1e6 if torch.is_tensor(out) else 0.0
            e = self.c[(name, site)]
            e[0] += 1; e[1] += mb
        return out


dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=40)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(3):
    tr.step(s0, s1)
which = sys.argv[1] if len(sys.argv) > 1 else "slim"
if which == "glue":  # the eager launches AROUND the replayed graphs of steady-state loop steps (2 pairs per step, 4 steps)
    pairs = [slim_pair(10 + i % 2, dev, n_points=120000, grid=512, bev_range_m=100.0) for i in range(8)]  # (two distinct pairs: one graph signature)
    ring = lambda i: [pairs[(2 * i + k) % 8] for k in range(2)]  # noqa: E731
    up = lambda i: [p for j in range(1, 5) for p in ring(i + j)]  # noqa: E731
    for i in range(16):
        tr.step_batch(ring(i), upcoming=up(i))
    torch.cuda.synchronize()
    with Sites() as st:
        for i in range(16, 20):
            tr.step_batch(ring(i), upcoming=up(i))
    tot = sum(v[0] for v in st.c.values())
    print(f"glue: {tot / 4:.1f} non-view aten ops per step outside the graphs")
    for (name, site), (n, mb) in sorted(st.c.items(), key=lambda kv: -kv[1][0])[:80]:
        print(f"{n / 4:6.1f} x {name:22s} {mb / 4:9.2f} MB  {site}")
    sys.exit(0)
with Sites() as st:
    if which == "slim":
        with torch.no_grad():
            tr.slim.infer_point_flow_t0_t1(s0, s1)
    elif which == "mine":
        with torch.no_grad():
            flow = tr.slim.infer_point_flow_t0_t1(s0, s1)
        st.c.clear()
        tr._targets_from_flow(s0, flow)
    else:
        boxes, _ = tr.mine_boxes(s0, s1)
        targets, _ = tr._targets_from_flow(s0, tr.slim.infer_point_flow_t0_t1(s0, s1))
        st.c.clear()
        tr.detector.eager_pass(s0["pcl_full_no_ground_ta"], targets)
tot = sum(v[0] for v in st.c.values())
print(f"{which}: {tot} non-view aten ops")
for (name, site), (n, mb) in sorted(st.c.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{n:4d} x {name:22s} {mb:9.2f} MB  {site}")
