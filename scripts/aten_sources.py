"""Where do the framework's own launches (ATen element-wise / cat / fill / copy / sort kernels) of one workload come from?

  python scripts/aten_sources.py loop|slim|detector  -> table: python frame inside liso_amd/ | op | launches | device us

One EAGER pass of the workload under a TorchDispatchMode: every ATen op that is not a view / allocation is charged to the innermost
frame of this repository on the python stack (backward ops run on the autograd thread: their stack starts in the custom
Function's backward, or is empty for built-in nodes -> "(autograd)").  The own HIP kernels are launched through ctypes and do not
appear as ops -- this table is exactly the part of the step that is NOT own code, i.e. the work list of "write producers straight
into consumers' slices"."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "loop"
GRAPH = what.endswith("-graph")  # the pipelined / captured step instead of the eager pass: ops inside captures + eager ops of a late step
what = what.replace("-graph", "")
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = default_cfg(grid=512, bev_range_m=100.0)
if what == "loop":
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.trainer import LisoLoopTrainer

    cfg = apply_slim_simple_knn_training(cfg)
    tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=64, use_graph=GRAPH, overlap=GRAPH, infer_batch=4, flow_ahead=2)
    pairs = [slim_pair(2 + 100 * i, dev, n_points=120000, grid=512, bev_range_m=100.0) for i in range(16 if GRAPH else 2)]
    if GRAPH:
        ctr = [0]

        def run():
            i = ctr[0] * 2
            ctr[0] += 1
            return tr.step_batch([pairs[(i + k) % 16] for k in range(2)], upcoming=tuple(pairs[(i + k) % 16] for k in range(2, 2 + 11)))
    else:
        run = lambda: tr.eager_pass_batch(pairs)  # noqa: E731
elif what == "slim":
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.trainer import SlimTrainer

    cfg = apply_slim_simple_knn_training(cfg)
    tr = SlimTrainer(cfg, dev, use_graph=False)
    s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
    run = lambda: tr.step(s0, s1, eager=True)  # noqa: E731
else:
    from liso_amd.datasets.synthetic import detector_batch
    from liso_amd.trainer import DetectorTrainer

    tr = DetectorTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=64, use_graph=GRAPH)
    pcls, targets = detector_batch(seed=1, batch=4, device=dev, n_points=120000, grid=512, bev_range_m=100.0)
    run = (lambda: tr.step(pcls, targets)) if GRAPH else (lambda: tr.eager_pass(pcls, targets))  # noqa: E731

if not GRAPH:
    for _ in range(2):
        run()
torch.cuda.synchronize()
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_KERNEL = ("empty", "new_empty", "sym_", "_local_scalar_dense", "is_", "_to_copy_meta", "lift_fresh", "detach", "alias", "resize_",
             "set_", "record_stream", "_unsafe_view", "_reshape_alias", "is_pinned", "_pin_memory")
agg = collections.Counter()


def storages(args):
    out = set()
    for a in args:
        if isinstance(a, torch.Tensor):
            out.add(a.untyped_storage().data_ptr())
        elif isinstance(a, (list, tuple)):
            out |= storages(a)
    return out


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0] if hasattr(func, "__name__") else str(func)
        if any(name.startswith(p_) for p_ in NO_KERNEL):
            return out
        outs = storages(out if isinstance(out, (list, tuple)) else [out])
        inplace = name.endswith("_") or (kwargs or {}).get("out") is not None
        if not inplace and outs and outs <= storages(args):  # a view of an input: no launch
            return out
        if not any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()) +
                   (list(out) if isinstance(out, (list, tuple)) else [out])):
            return out
        frame = "(autograd)"
        for fs in reversed(traceback.extract_stack()[:-1]):
            if "/liso_amd/" in fs.filename or fs.filename.endswith("bench.py"):
                frame = f"{fs.filename.split(ROOT + '/')[-1]}:{fs.lineno} {fs.name}"
                break
        if GRAPH:
            if torch.cuda.is_current_stream_capturing():
                frame = "[in graph] " + frame
            elif STEP[0] < LAST:
                return out  # (eager ops of the early steps: warm-up passes in front of the captures)
        agg[(frame, name)] += 1
        return out


STEP, LAST = [0], 11
with Count():
    if GRAPH:  # every capture happens within the first steps; the eager-side ops are those of the last one
        for k in range(LAST + 1):
            STEP[0] = k
            run()
    else:
        run()
torch.cuda.synchronize()
print(f"# {what}: {sum(agg.values())} framework ops that launch kernels in one eager pass")
by_frame = collections.defaultdict(lambda: [0, []])
for (frame, name), n in agg.items():
    by_frame[frame][0] += n
    by_frame[frame][1].append(f"{name} x{n}")
by_file = collections.Counter()
for frame, (n, names) in by_frame.items():
    by_file[frame.split(":")[0]] += n
print("# per file:", ", ".join(f"{k} {v}" for k, v in by_file.most_common()))
for frame, (n, names) in sorted(by_frame.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:5d}  {frame[:100]:100s} {', '.join(sorted(names))[:200]}")
