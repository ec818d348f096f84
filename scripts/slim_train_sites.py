"""which source lines issue the ATen launches of one eager SLIM training step (forward + backward)?
(op, innermost liso_amd frame, output shape) -> count, output MB.  Backward ops issued by the autograd engine have no python frame
of ours ("?"): their shapes tell which forward op they belong to."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from liso_amd.utils.config import default_cfg, apply_slim_simple_knn_training
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import SlimTrainer

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
            shape = tuple(out.shape) if torch.is_tensor(out) else ()
            mb = out.numel() * out.element_size() / 1e6 if torch.is_tensor(out) else 0.0
            e = self.c[(name, site, shape)]
            e[0] += 1; e[1] += mb
        return out


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    cfg = default_cfg(grid=512, bev_range_m=100.0)
    if "full" not in sys.argv[1:]:  # `full`: the reference's default SLIM losses (static-flow + fw/bw transform penalties) instead of the kNN-only overlay
        cfg = apply_slim_simple_knn_training(cfg)
    torch.manual_seed(0)
    tr = SlimTrainer(cfg, dev, use_graph=False)
    s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
    for _ in range(2):
        tr.step(s0, s1, eager=True, update=False)
    with Sites() as st:
        tr.step(s0, s1, eager=True, update=False)
    tot = sum(v[0] for v in st.c.values())
    print(f"{tot} non-view aten ops")
    by_op = collections.Counter()
    for (name, site, shape), (n, mb) in st.c.items():
        by_op[name] += n
    print(by_op.most_common(25))
    by_file = collections.Counter()
    for (name, site, shape), (n, mb) in st.c.items():
        by_file[site.split(":")[0]] += n
    print(by_file.most_common())
    for (name, site, shape), (n, mb) in sorted(st.c.items(), key=lambda kv: -kv[1][0])[:400]:
        print(f"{n:4d} x {name:20s} {mb:9.2f} MB {str(shape):28s} {site}")
