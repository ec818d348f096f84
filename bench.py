#!/usr/bin/env python3
"""bench.py -- headline benchmark of the LISO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one CenterPoint-pillar detector TRAIN step (voxelise -> fused PFN/scatter -> BEV backbone -> CenterHead ->
decode -> loss -> backward -> [RCCL gradient all-reduce] -> AdamW -> OneCycleLR) over one batch of synthetic
KITTI-shaped 120k-point clouds already resident in HBM (BASELINE.json configs[2]; see DESIGN.md "Measurement" for
why this is the N=1 workload of this round).  Weak scaling: per-GPU batch fixed, samples sharded across ranks, the
only collective is the gradient all-reduce.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_POINTS = 120000
GRID = 512
BEV_RANGE = 100.0
BATCH_PER_GPU = 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def pfn_algorithmic_bytes(batch, n_points, grid, out_bytes):
    """SURVEY.md 8(d), pillar path: read the points once (N*C*4 B) + write the dense canvas once (64*G^2*s B) +
    occupancy (G^2*4 B), per sample.  The launch timed here is the fused PFN+scatter kernel; the canvas zero-fill
    that precedes it is part of the same algorithmic write and is NOT counted twice."""
    return batch * (n_points * 4 * 4 + 64 * grid * grid * out_bytes + grid * grid * 4)


def cpu_baseline(trainer, pcls, targets):
    """The oracle port of the same train step (fwd+bwd) on the host cores, ONE frame (bounded sample)."""
    from oracle.train_step import timed_detector_step

    sd = {k: v.detach().float().cpu() for k, v in trainer.net.state_dict().items()}
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    t1 = {k: v[:1].cpu() for k, v in targets.items()}
    secs, _ = timed_detector_step(sd, [pcls[0].cpu()], t1, GRID, BEV_RANGE)
    return {"value": 1.0 / secs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 frame ({N_POINTS} pts, {GRID}x{GRID} BEV) fwd+bwd, fp32, torch-CPU oracle, {secs:.2f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the HIP path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from liso_amd import _lib as L
    from liso_amd.datasets.synthetic import detector_batch
    from liso_amd.trainer import DetectorTrainer
    from liso_amd.utils.config import default_cfg

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = default_cfg(grid=GRID, bev_range_m=BEV_RANGE)
    torch.manual_seed(0)  # identical initial weights on every rank (DDP also broadcasts them)
    trainer = DetectorTrainer(cfg, dev, compute_dtype=dtype, total_steps=args.steps + args.warmup + 8)
    # each rank owns different samples (DistributedSampler-style sharding by seed), resident in HBM
    pcls, targets = detector_batch(seed=1 + rank, batch=args.batch, device=dev, n_points=N_POINTS, grid=GRID,
                                   bev_range_m=BEV_RANGE)

    for _ in range(args.warmup):
        trainer.step(pcls, targets)

    L.TIMER.enable("pfn_forward_scatter")
    L.TIMER.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(pcls, targets)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.TIMER.disable_all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        frames = args.batch * world * args.steps
        durs = L.TIMER.durations_ms("pfn_forward_scatter")
        avg_ms = sum(durs) / max(len(durs), 1)
        out_bytes = 2 if dtype == torch.bfloat16 else 4
        alg = pfn_algorithmic_bytes(args.batch, N_POINTS, GRID, out_bytes)
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        line = {
            "metric": "LISO train-step frames/sec (120k-pt clouds)",
            "value": frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "CenterPoint-pillar detector train step (BASELINE configs[2]): 120k-pt KITTI-shaped "
                                   "clouds, 512x512 BEV pillars, fwd+bwd+AdamW",
                       "points_per_cloud": N_POINTS, "bev_grid": GRID, "batch_per_gpu": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "final_loss": float(loss),
            "roofline": {"kernel": "pfn_forward_kernel (fused decorate+Linear+BN+ReLU+max+scatter)", "bound": "hbm",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(trainer, pcls, targets)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
