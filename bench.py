#!/usr/bin/env python3
"""bench.py -- headline benchmark of the LISO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default workload (`--workload slim`, BASELINE.json configs[1], the configuration the metric is quoted on): a "step" =
one SLIM self-supervised TRAIN step (pillar-encode both clouds -> RAFT fwd+bw, 6 iterations each -> flow/class
decoder with weighted Kabsch -> kNN loss over all 6 iterations -> backward -> [RCCL gradient all-reduce] -> RMSprop)
on one pair of synthetic KITTI-shaped 120k-point clouds already resident in HBM, B=1 per GPU as in the reference's
`slim_RAFT batch_size_one`; a step consumes 2 frames.
`--workload detector` (configs[2]): one CenterPoint-pillar detector train step (voxelise -> fused PFN/scatter -> BEV
backbone -> CenterHead -> decode -> loss -> backward -> [all-reduce] -> AdamW -> OneCycleLR), B=4 clouds per GPU.
`--workload loop` (configs[3]): the fused LISO iteration -- SLIM forward (no_grad) -> flow clusters (DBSCAN) -> NMS ->
target maps -> detector train step, one sweep pair per GPU.
Weak scaling: per-GPU work fixed, samples sharded across ranks by seed, the only collective is the gradient
all-reduce.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _seed_miopen_user_db():
    """MIOpen keeps the solver picks / tuning parameters it finds in a per-user database; a fresh machine starts from
    heuristics (detector step 8.9 ms) and only reaches the tuned state (6.0-6.6 ms) after several processes have run.
    liso_amd/miopen_db/ holds that database as harvested on an MI355X with this ROCm image; every process works on a
    private copy (MIOpen appends to it).  Must run before MIOpen initialises; no effect if the files do not match the
    installed MIOpen version."""
    src = os.path.join(ROOT, "liso_amd", "miopen_db")
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(src) or "--no-miopen-db" in sys.argv:
        return
    import shutil
    import tempfile
    dst = tempfile.mkdtemp(prefix="liso_miopen_db_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst


_seed_miopen_user_db()

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
    ap.add_argument("--workload", default="slim", choices=["slim", "detector", "loop"])
    ap.add_argument("--batch", type=int, default=None, help="detector workload: clouds per GPU (default 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="slim workload: replay forward+loss+backward from a hipGraph (immune to host jitter: a steady 24.5 ms per "
                         "step; eager launches are 22-23 ms on an idle host and up to 30 ms on a busy one, and only eager steps "
                         "can carry the per-kernel HIP events inside the timed region, so eager is the default)")
    ap.add_argument("--conv-benchmark", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: MIOpen re-times its solvers in this process (run-to-run "
                         "variation of the picks: 540-620 frames/s on the detector); default: the picks of the seeded database")
    ap.add_argument("--no-miopen-db", action="store_true", help="do not seed MIOpen's user database from liso_amd/miopen_db/")
    ap.add_argument("--nhwc", action="store_true", help="slim workload: conv filters in channels-last memory format (slower)")
    return ap.parse_args()


def pfn_algorithmic_bytes(batch, n_points, grid, out_bytes):
    """SURVEY.md 8(d), pillar path: read the points once (N*C*4 B) + write the dense canvas once (64*G^2*s B) +
    occupancy (G^2*4 B), per sample."""
    return batch * (n_points * 4 * 4 + 64 * grid * grid * out_bytes + grid * grid * 4)


def slim_algorithmic_bytes(batch, n_points, grid, levels=4, radius=3, directions=2):
    """SURVEY.md 8(d), SLIM rows, per launch:
    corr lookup (fwd, and its adjoint bwd), per sample: levels * hw * (2r+1)^2 bilinear reads of 4 taps * 4 B + the
      [hw, levels*(2r+1)^2] fp32 output, hw = (G/8)^2;
    1-NN query: (N_q + N_ref) * 12 B in + N_q * 8 B out."""
    hw = (grid // 8) ** 2
    w2 = (2 * radius + 1) ** 2
    # training: the forward and the backward flow direction of every pair share one launch (2 * batch samples per lookup);
    # the box miner of the loop workload runs the forward direction only
    lookup = directions * batch * (levels * hw * w2 * 4 * 4 + hw * levels * w2 * 4)
    # RAFT output assembly: [6 iterations x directions x batch, G, G, 8] fp32 written (fwd) / read (bwd) once + the low-res maps
    outputs = 6 * directions * batch * (grid * grid * 8 * 4 + hw * 6 * 4)
    return {"corr_lookup_fwd": lookup, "corr_lookup_bwd": lookup, "knn_query": 2 * n_points * 12 + n_points * 8,
            "raft_outputs_fwd": outputs, "raft_outputs_bwd": outputs}


SLIM_KERNELS = {
    "corr_lookup_fwd": "corr_lookup_fwd_kernel (on-the-fly 4-level correlation + bilinear lookup)",
    "corr_lookup_bwd": "corr_lookup_bwd_kernel (adjoint of the lookup into fmap1 / pooled fmap2 gradients)",
    "knn_query": "knn_query_kernel (exact 1-NN, two-level bucket grid with z bins, 16 lanes per query, one launch)",
    "raft_outputs_fwd": "upsample_fwd_kernel (x8 bilinear upsampling + flow convention + concat of all RAFT iterations)",
    "raft_outputs_bwd": "upsample_bwd_x/y kernels (adjoint of the output assembly, two gather passes)",
}


LOOP_KERNELS = {  # name -> (description, algorithmic bytes per launch at B=1, N=120k, G=512)
    "pfn_forward_scatter": ("pfn_forward_kernel (Linear+BN+ReLU+max + dense scatter from CSR feature rows)", None),
    "corr_lookup_fwd": (SLIM_KERNELS["corr_lookup_fwd"], None),
    "bev_dynamic_flow": ("bev_scatter_kernel + bev_mean_kernel (non-rigid flow, fixed-point scatter-mean)",
                         N_POINTS * (12 + 12 + 8 + 1) + GRID * GRID * 16),
    "dbscan_components": ("dbscan core/union/flatten kernels (grid-window DBSCAN)", GRID * GRID * (1 + 12 + 1 + 4 + 4)),
    "dbscan_labels": ("dbscan_label_kernel", GRID * GRID * (1 + 12 + 1 + 4 + 4 + 4)),
    "kabsch_trafos": ("kabsch_moments_kernel + kabsch_solve_kernel (soft masks + weighted Kabsch)", N_POINTS * (12 + 8 + 1)),
    "fit_box_z": ("fit_z_kernel (points-in-box z extent)", N_POINTS * 12),
}


def pmc_traffic(workload, patterns):
    """HBM bytes per launch of the roofline kernel(s) from the committed rocprofv3 PMC passes of this bench command
    (profiles/r01_<workload>_pmc_{FETCH,WRITE}_SIZE.csv; counters cannot be collected from inside the timed process).
    MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide streaming
    reads, so reads are doubled (an upper bracket for the narrow / scattered reads of these kernels)."""
    import csv

    total = 0.0
    for pat in patterns:
        for kind, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            path = os.path.join(ROOT, "profiles", f"r01_{workload}_pmc_{kind}.csv")
            if not os.path.exists(path):
                return None
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
            if not vals:
                return None
            total += factor * 1024.0 * sum(vals) / len(vals)
    return total


def cpu_baseline_detector(trainer, pcls, targets):
    """The oracle port of the same train step (fwd+bwd) on the host cores, ONE frame (bounded sample)."""
    from oracle.train_step import timed_detector_step

    sd = {k: v.detach().float().cpu() for k, v in trainer.net.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t1 = {k: v[:1].cpu() for k, v in targets.items()}
    secs, _ = timed_detector_step(sd, [pcls[0].cpu()], t1, GRID, BEV_RANGE)
    return {"value": 1.0 / secs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 frame ({N_POINTS} pts, {GRID}x{GRID} BEV) fwd+bwd, fp32, torch-CPU oracle, {secs:.2f} s"}


def cpu_baseline_slim(cfg, trainer, s0, s1):
    """The CPU port of the same SLIM step (oracle/slim_step.py) on the host cores: ONE pair = 2 frames."""
    from oracle.slim_step import timed_slim_step

    sd = {k: v.detach().cpu() for k, v in trainer.net.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    secs, _ = timed_slim_step(cfg, sd, s0, s1)
    return {"value": 2.0 / secs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 pair = 2 frames ({N_POINTS} pts each, {GRID}x{GRID} BEV), 1 SLIM train step fwd+bwd+RMSprop, "
                      f"fp32, torch-CPU port with explicit correlation volume + cKDTree, {secs:.2f} s"}


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
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

    torch.backends.cudnn.benchmark = bool(args.conv_benchmark)

    cfg = default_cfg(grid=GRID, bev_range_m=BEV_RANGE)
    torch.manual_seed(0)  # identical initial weights on every rank (DDP also broadcasts them)
    if args.workload == "slim":
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import SlimTrainer

        args.dtype = "fp32"  # the reference trains SLIM in fp32 (no autocast in slim/experiment.py)
        batch = 1  # one pair per GPU, as in the reference's `slim_RAFT batch_size_one`
        cfg = apply_slim_simple_knn_training(cfg)
        trainer = SlimTrainer(cfg, dev, use_graph=args.graph, channels_last=args.nhwc)
        # each rank owns different pairs (DistributedSampler-style sharding by seed), resident in HBM
        s0, s1 = slim_pair(2 + rank, dev, n_points=N_POINTS, grid=GRID, bev_range_m=BEV_RANGE)
        step = lambda: trainer.step(s0, s1)  # noqa: E731
        frames_per_step, timed = 2 * batch, list(SLIM_KERNELS)
    elif args.workload == "loop":
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import LisoLoopTrainer

        batch = 1
        dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        cfg = apply_slim_simple_knn_training(cfg)
        trainer = LisoLoopTrainer(cfg, dev, compute_dtype=dtype, total_steps=args.steps + args.warmup + 8)
        s0, s1 = slim_pair(2 + rank, dev, n_points=N_POINTS, grid=GRID, bev_range_m=BEV_RANGE)
        step = lambda: trainer.step(s0, s1)  # noqa: E731
        frames_per_step, timed = 2, list(LOOP_KERNELS)
    else:
        from liso_amd.datasets.synthetic import detector_batch
        from liso_amd.trainer import DetectorTrainer

        batch = args.batch or BATCH_PER_GPU
        dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        trainer = DetectorTrainer(cfg, dev, compute_dtype=dtype, total_steps=args.steps + args.warmup + 8)
        pcls, targets = detector_batch(seed=1 + rank, batch=batch, device=dev, n_points=N_POINTS, grid=GRID,
                                       bev_range_m=BEV_RANGE)
        step = lambda: trainer.step(pcls, targets)  # noqa: E731
        frames_per_step, timed = batch, ["pfn_decorate", "pfn_forward_scatter"]

    graph_note = None
    if args.workload == "slim" and args.graph:
        # capture before the first step and let all ranks agree on the outcome (capture issues no collective)
        ok, err = 1, ""
        try:
            trainer.capture(s0, s1)
        except Exception as e:  # capture refused by the runtime: same kernels, launched eagerly
            ok, err = 0, f"{type(e).__name__}: {str(e)[:120]}"
        if world > 1:
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            graph_note = f"hipGraph capture failed on some rank ({err}); eager launches"
            print(graph_note, file=sys.stderr, flush=True)
            args.graph = False
            torch.manual_seed(0)
            trainer = SlimTrainer(cfg, dev, use_graph=False, channels_last=args.nhwc)
            step = lambda: trainer.step(s0, s1)  # noqa: E731
    for _ in range(args.warmup):
        step()

    graphed = args.workload == "slim" and args.graph
    if not graphed:  # per-launch HIP events on the launch stream, inside the timed region
        for k in timed:
            L.TIMER.enable(k)
    L.TIMER.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timed_in = "the timed steps"
    if graphed and rank == 0:
        # the timed steps replay a hipGraph, inside which per-kernel events cannot be recorded: the same kernels are
        # timed over two eager steps on the same inputs right after the timed region (rocprof of the graph replays
        # agrees, profiles/)
        for k in timed:
            L.TIMER.enable(k)
        L.TIMER.reset()
        n_event_steps = 2
        for _ in range(n_event_steps):
            trainer.step(s0, s1, eager=True)
        torch.cuda.synchronize()
        timed_in = f"{n_event_steps} eager steps after the timed region (the timed steps replay a hipGraph)"
    L.TIMER.disable_all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        durs = {k: L.TIMER.durations_ms(k) for k in timed}
        event_steps = n_event_steps if graphed else args.steps
        if args.workload == "loop":
            key = max(durs, key=lambda k: sum(durs[k]))
            alg_all = dict(slim_algorithmic_bytes(1, N_POINTS, GRID, directions=1))
            alg_all["pfn_forward_scatter"] = pfn_algorithmic_bytes(1, N_POINTS, GRID, 4)
            alg = LOOP_KERNELS[key][1] or alg_all[key]
            kname = LOOP_KERNELS[key][0]
            workload = ("fused LISO iteration (BASELINE configs[3]): SLIM fwd (no_grad) -> FlowClusterDetector (DBSCAN) -> NMS "
                        "-> target maps -> CenterPoint-pillar train step, one 120k-pt sweep pair per GPU, 512x512 BEV")
        elif args.workload == "slim":
            alg_all = slim_algorithmic_bytes(batch, N_POINTS, GRID)
            key = max(durs, key=lambda k: sum(durs[k]))  # the hand-written kernel with the largest share of the step
            alg, kname = alg_all[key], SLIM_KERNELS[key]
            if key == "knn_query":  # queries per launch as counted at the call site
                nq = L.TIMER.mean_units("knn_query") or N_POINTS
                alg = int((nq + N_POINTS) * 12 + nq * 8)
            workload = ("SLIM scene-flow train step (BASELINE configs[1]): two 120k-pt KITTI-shaped clouds, 512x512 BEV "
                        "pillars, RAFT 6 iterations fwd+bw flow, kNN loss, fwd+bwd+RMSprop")
        else:
            key = "pfn_forward_scatter"
            alg = pfn_algorithmic_bytes(batch, N_POINTS, GRID, 2 if args.dtype == "bf16" else 4)
            kname = ("pfn_decorate_kernel (+3 scan kernels) then pfn_forward_kernel: decorate -> Linear+BN+ReLU+max -> dense "
                     "scatter; the two launches are timed together")
            workload = ("CenterPoint-pillar detector train step (BASELINE configs[2]): 120k-pt KITTI-shaped clouds, "
                        "512x512 BEV pillars, fwd+bwd+AdamW")
        pmc_patterns = {"slim": {"knn_query": ["knn_query_kernel"], "corr_lookup_fwd": ["corr_lookup_fwd_kernel"]}.get(key),
                        "detector": ["pfn_decorate_kernel", "pfn_forward_kernel"],
                        "loop": {"corr_lookup_fwd": ["corr_lookup_fwd_kernel"],
                                 "dbscan_components": ["dbscan_core_kernel", "dbscan_union_kernel", "dbscan_flatten_kernel"]}.get(key)}[args.workload]
        traffic = pmc_traffic(args.workload, pmc_patterns) if pmc_patterns else None
        avg_ms = sum(durs[key]) / max(len(durs[key]), 1)
        if args.workload == "detector":  # the pillar pass is two launches (decorate, forward): time them as one unit
            avg_ms += sum(durs["pfn_decorate"]) / max(len(durs["pfn_decorate"]), 1)
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        line = {
            "metric": "LISO train-step frames/sec (120k-pt clouds)",
            "value": frames_per_step * world * args.steps / elapsed,
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
            "config": {"workload": workload, "points_per_cloud": N_POINTS, "bev_grid": GRID, "batch_per_gpu": batch,
                       "frames_per_step_per_gpu": frames_per_step, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay of fwd+loss+bwd, eager RMSprop" if graphed else "eager",
                       "miopen_solver_selection": ("timed in-process (cudnn.benchmark)" if args.conv_benchmark else
                                                   "MIOpen defaults" if args.no_miopen_db else "seeded user database liso_amd/miopen_db")},
            "final_loss": float(loss),
            "roofline": {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_source": None if traffic is None else f"profiles/r01_{args.workload}_pmc_*.csv (separate rocprofv3 --pmc "
                                                                        "passes of this command; 2 x FETCH_SIZE + WRITE_SIZE, KiB)",
                         "avg_launch_ms": avg_ms,
                         "launches_per_step": len(durs[key]) / max(event_steps, 1), "algorithmic_bytes_per_launch": alg,
                         "timed_in": timed_in,
                         "timed_kernels_ms_per_step": {k: sum(v) / max(event_steps, 1) for k, v in durs.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "slim":
                line["cpu_baseline"] = cpu_baseline_slim(cfg, trainer, s0, s1)
            elif args.workload == "detector":
                line["cpu_baseline"] = cpu_baseline_detector(trainer, pcls, targets)
        if args.workload == "loop":
            line["mined_boxes_last_step"] = int(trainer.last_boxes.valid.sum())
        if graph_note:
            line["config"]["launch"] = graph_note
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
