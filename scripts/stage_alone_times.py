"""GPU time of the stages of the LISO loop when each runs alone, at the bench's batch sizes:
A = SLIM inference of `--ib` pairs per replay (pillars eager + graph), B = box mining of one pair (graph), C = detector step on `--batch` pairs"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer, _BatchedTargets
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

arg = lambda k, d: int(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d  # noqa: E731
NB, IB = arg("--batch", 2), arg("--ib", 4)
dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=400, use_graph=True, overlap=True, infer_batch=IB, flow_ahead=2)
pairs = [slim_pair(2 + 100 * i, dev) for i in range(max(IB, NB) + 1)]
N = 40


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(N):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / N


with torch.no_grad():
    s0 = tr._stack_samples([p[0] for p in pairs[:IB]]); s1 = tr._stack_samples([p[1] for p in pairs[:IB]])
    flow = tr._infer_flow(s0, s1)
    b = flow.shape[0] // IB
    f0 = flow[:b].clone()
    a_ms = timed(lambda i: tr._infer_flow(s0, s1))
    print(f"A alone, {IB} pairs per replay: {a_ms:.2f} ms = {a_ms / IB:.2f} ms per pair")
    g = list(tr._infer_graphs.values())[-1]["graph"]
    ag = timed(lambda i: g.replay())
    print(f"A graph replay only: {ag:.2f} ms = {ag / IB:.2f} per pair (eager pillar encoders + copies: {(a_ms - ag) / IB:.2f} per pair)")
    side = tr._mine_stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr._mine_from_graph(pairs[0][0], f0, side)
        b_ms = timed(lambda i: tr._mine_from_graph(pairs[i % 2][0], f0, side))
    torch.cuda.current_stream().wait_stream(side)
    print(f"B alone (graph, one pair): {b_ms:.2f} ms")
    per = [tr._targets_from_flow(p[0], f0, capacity=tr.box_capacity)[0] for p in pairs[:NB]]
targets = per[0] if NB == 1 else _BatchedTargets(per)
pcls = [c for p in pairs[:NB] for c in p[0]["pcl_full_no_ground_ta"]]
c_ms = timed(lambda i: tr.detector.step(pcls, targets))
print(f"C alone (detector step on {NB} pairs): {c_ms:.2f} ms = {c_ms / NB:.2f} per pair")
gd = tr.detector._graph
cg = timed(lambda i: gd.replay())
print(f"C graph replay only: {cg:.2f} ms (eager pillar encoder fwd+bwd, AdamW, copies: {c_ms - cg:.2f})")
print(f"sum per step of {NB} pairs: A {NB * a_ms / IB:.2f} + B {NB * b_ms:.2f} + C {c_ms:.2f} = {NB * a_ms / IB + NB * b_ms + c_ms:.2f} ms")
