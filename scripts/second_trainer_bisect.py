"""Why does a LisoLoopTrainer that is built SECOND in a process step at ~6.6 ms instead of ~4.35 (scripts/two_trainers.py)?
One experiment per child process (python scripts/second_trainer_bisect.py runs them all; `... <mode>` runs one):
  base          trainer alone (reference number)
  built_only    a first trainer is BUILT, never stepped, deleted; then the measured one
  stepped       a first trainer is built and stepped 12 times, kept alive (the known slow case)
  seq           as `stepped`, but the measured trainer runs WITHOUT overlap (one stream): is it the kernels or the concurrency?
  seq_base      a lone trainer without overlap (reference for `seq`)
  detector1st   a DetectorTrainer (stepped) first, then the loop trainer
  slim1st       a SlimTrainer (stepped, eager) first, then the loop trainer: the real training process
  sync_streams  as `stepped`, first trainer deleted, gc + empty_cache + torch.cuda.synchronize, and its side streams' pools untouched
  newpool       as `stepped`, the measured trainer's side streams taken as HIGH-priority streams (a different stream pool)
"""
import gc
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = ["base", "built_only", "stepped", "seq_base", "seq", "detector1st", "slim1st", "sync_streams", "newpool"]


def main(mode):
    import torch

    from liso_amd.datasets.synthetic import detector_batch, slim_pair
    from liso_amd.trainer import DetectorTrainer, LisoLoopTrainer, SlimTrainer
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

    dev = torch.device("cuda:0")
    cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
    pairs = [slim_pair(2 + 100 * i, dev, n_points=120000 + (i % 5 - 2) * 1500, grid=512, bev_range_m=100.0) for i in range(16)]
    batch, n_up = 2, 11

    def make(overlap=True):
        torch.manual_seed(0)
        return LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=512, use_graph=True, overlap=overlap, infer_batch=4, flow_ahead=2)

    def run(tr, steps, ctr):
        for _ in range(steps):
            i = ctr[0] * batch
            ctr[0] += 1
            tr.step_batch([pairs[(i + k) % 16] for k in range(batch)], upcoming=tuple(pairs[(i + k) % 16] for k in range(batch, batch + n_up)))

    def timed(tr, label):
        ctr = [0]
        run(tr, 12, ctr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(tr, 24, ctr)
        torch.cuda.synchronize()
        print(f"{mode:13s} {label}: {1e3 * (time.perf_counter() - t0) / 24:.3f} ms per step", flush=True)

    keep = None
    if mode == "built_only":
        first = make()
        del first
        gc.collect()
    elif mode in ("stepped", "seq", "newpool"):
        keep = make()
        run(keep, 12, [0])
        torch.cuda.synchronize()
    elif mode == "sync_streams":
        first = make()
        run(first, 12, [0])
        torch.cuda.synchronize()
        del first
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
    elif mode == "detector1st":
        keep = DetectorTrainer(default_cfg(grid=512, bev_range_m=100.0), dev, compute_dtype=torch.bfloat16, total_steps=64, use_graph=True)
        pcls, targets = detector_batch(1, 2, dev, n_points=120000, grid=512, bev_range_m=100.0)
        for _ in range(6):
            keep.step(pcls, targets)
        torch.cuda.synchronize()
    elif mode == "slim1st":
        keep = SlimTrainer(cfg, dev, use_graph=False)
        for _ in range(3):
            keep.step(*pairs[0])
        torch.cuda.synchronize()
    tr = make(overlap=mode not in ("seq", "seq_base"))
    if mode == "newpool":
        tr._flow_stream = torch.cuda.Stream(device=dev, priority=-1)
        tr._mine_stream = torch.cuda.Stream(device=dev, priority=-1)
        if hasattr(tr, "_mine_streams"):
            tr._mine_streams = [tr._mine_stream] + [torch.cuda.Stream(device=dev, priority=-1) for _ in tr._mine_streams[1:]]
    timed(tr, "measured trainer")
    if keep is not None and isinstance(keep, LisoLoopTrainer):
        timed(keep, "the FIRST trainer afterwards")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(sys.argv[1])
    else:
        for m in MODES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), m], capture_output=True, text=True, timeout=600)
            out = [ln for ln in r.stdout.splitlines() if "ms per step" in ln]
            print("\n".join(out) if out else f"{m:13s} FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)
