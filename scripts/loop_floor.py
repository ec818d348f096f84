"""Where do the 0.4 ms between the detector step alone (2.75 ms) and the pipelined loop with BOTH side stages cached (3.15-3.17 ms) go?
Variants, one per child process: python scripts/loop_floor.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(mode):
    import torch

    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.trainer import LisoLoopTrainer
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

    dev = torch.device("cuda:0")
    cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
    pairs = [slim_pair(2 + 100 * i, dev, n_points=120000, grid=512, bev_range_m=100.0) for i in range(16)]
    batch, n_up = 2, 11
    torch.manual_seed(0)
    overlap = "seq" not in mode
    tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=512, use_graph=True, overlap=overlap, infer_batch=4, flow_ahead=2)
    ctr = [0]

    def steps(n):
        for _ in range(n):
            i = ctr[0] * batch
            ctr[0] += 1
            up = tuple(pairs[(i + k) % 16] for k in range(batch, batch + n_up)) if "noannounce" not in mode else ()
            tr.step_batch([pairs[(i + k) % 16] for k in range(batch)], upcoming=up)

    steps(12)
    torch.cuda.synchronize()
    real_b, real_a, real_t = tr._mine_from_graph, tr._infer_flow_padded, tr._targets_from_flow
    cb, ca, ct = {}, {}, {}
    tr._mine_from_graph = lambda s, f, side: cb.setdefault("r", real_b(s, f, side)) if "r" not in cb else cb["r"]

    def infer(s0, s1):
        key = s0["pcl_ta"]["pcl"].shape
        if key not in ca:
            ca[key] = real_a(s0, s1).clone()
        return ca[key]

    def tff(sample, flow, capacity=None, **k):
        if "r" not in ct:
            ct["r"] = real_t(sample, flow, capacity=capacity, **k) if capacity is not None else real_t(sample, flow, **k)
        return ct["r"]

    tr._infer_flow_padded = infer
    if not overlap or "noannounce" in mode:
        tr._targets_from_flow = tff  # (the unannounced / sequential path mines eagerly through this call)
    steps(12)
    torch.cuda.synchronize()
    import collections
    acc = collections.Counter()

    def wrap(obj, name):
        inner = getattr(obj, name)

        def f(*a, **k):
            t = time.perf_counter()
            try:
                return inner(*a, **k)
            finally:
                acc[name] += time.perf_counter() - t
        setattr(obj, name, f)

    if "timers" in mode:
        for n in ("_take_mined", "_stage_a", "_stage_b"):
            wrap(tr, n)
        wrap(tr.detector, "step")
    t0 = time.perf_counter()
    steps(40)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    if acc:
        print("   " + "  ".join(f"{k} {1e3 * v / 40:.3f}" for k, v in acc.items()), flush=True)
    print(f"{mode:28s} host {1e3 * host / 40:.3f} ms per step, GPU done after {1e3 * (time.perf_counter() - t0) / 40:.3f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(sys.argv[1])
    else:
        for mode, env in (("cached", {}), ("cached_noprep", {"LISO_PREP_AHEAD": "0"}), ("cached_seq", {}), ("cached_noannounce", {}),
                          ("cached_allcus", {"LISO_INFER_CUS": "0"}), ("cached_timers", {}), ("cached_noinputsready_timers", {"LISO_INPUTS_READY": "0"})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, **env))
            out = [ln for ln in r.stdout.splitlines() if "ms per step" in ln or ln.startswith("   ")]
            print("\n".join(out) if out else f"{mode} FAILED rc={r.returncode} {r.stderr[-500:]}", flush=True)
