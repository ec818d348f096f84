#!/usr/bin/env python3
"""bench.py -- headline benchmark of the LISO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default workload (`--workload loop`, BASELINE.json configs[3] at N GPUs, the workload north_star's "Target" sentence
defines the metric on): a "step" = one fused LISO iteration per GPU -- SLIM forward (no_grad) on a pair of 120k-point
sweeps -> per-point flow -> FlowClusterDetector (BEV dynamicness, DBSCAN, region moments, z fit, Kabsch heading) ->
rotated NMS -> CenterPoint target maps -> CenterPoint-pillar detector train step (fwd + bwd + [RCCL gradient all-reduce]
+ AdamW + OneCycleLR), inputs resident in HBM; a step consumes 2 frames.  The same JSON line carries `iou3d_nms`
(boxes/s of the NMS entry point incl. the device greedy sweep, pairs/s of the IoU matrix; the second half of
BASELINE.json's metric), `roofline` (the hand-written kernel family with the largest share of the step's GPU time,
HIP events on the launch stream) and `cpu_baseline` (the oracle port of the same iteration on the host cores).
`--workload slim` (configs[1]): one SLIM self-supervised TRAIN step on one pair, B=1 per GPU as in the reference's
`slim_RAFT batch_size_one`.  `--workload detector` (configs[2]): one detector train step, B=4 clouds per GPU, bf16.
`--workload iou3d`: only the iou3d_nms section.  `--workload stress` (configs[4] on one GPU): the detector train step on
nuScenes-shaped 300k-point clouds (5 channels: x, y, z, intensity, time), 1024 x 1024 BEV, bf16 -- the HBM-bound stress of
the pillar path; the line carries `roofline_pillars` (decorate + forward launches timed together, HBM roofline) beside the
dominant kernel's `roofline`.
`--dtype`: bf16 (default: detector in bf16 as BASELINE configs[2] says, SLIM in fp32 tensors on F32X3 MFMAs) | f32x3 (fp32 tensors
everywhere, three bf16 MFMAs per product: the cheapest arithmetic that meets north_star's 1e-3 on logits and flow,
tests/test_gpu_parity_full_size.py) | fp32 (native fp32 MFMA everywhere).  The default line also carries bounded legs of the same
iteration in the two parity-conformant arithmetics (`parity_leg`, `fp32_exact_leg`) and of configs[1] / [2] / [4]
(`slim_leg`, `detector_leg`, `stress_leg`: `python bench.py --workload ...` run as child processes, their own lines trimmed).
Weak scaling: per-GPU work fixed, samples sharded across ranks by seed, the only collective is the gradient
all-reduce.  Rank 0 prints ONE JSON line.  `python bench.py --gpus N` without a launcher starts the N ranks itself.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _graph_env():
    """`--workload slim --graph` captures autograd's reductions / rocPRIM scans, whose hipMemsetAsync nodes the runtime only
    replays correctly with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (liso_amd/utils/graph_safety.py); the loop / detector graphs hold
    no memset node and keep the default (recorded packets: 1 ms per step faster).  Must be set before HIP initialises."""
    if "--graph" in sys.argv and "slim" in sys.argv:
        os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


_graph_env()

N_POINTS = 120000
GRID = 512
BEV_RANGE = 100.0
BATCH_PER_GPU = 4
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_BF16_TF = 2500.0  # dense bf16 MFMA
VALU_PEAK_F32_TF = 157.3    # fp32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32x3", "fp32"])
    ap.add_argument("--workload", default="loop", choices=["loop", "slim", "detector", "iou3d", "stress"])
    ap.add_argument("--lookahead", type=int, default=7,
                    help="loop workload: sweep pairs announced ahead of the current one (stage A infers lookahead - 1 - flow_ahead pairs per replay)")
    ap.add_argument("--flow-ahead", type=int, default=2,
                    help="loop workload: steps by which a SLIM inference batch is issued before stage B needs its first flow")
    ap.add_argument("--no-overlap", action="store_true",
                    help="loop workload: all stages of an iteration on one stream, one pair at a time (default: SLIM inference "
                         "of pair i+2 and box mining of pair i+1 on their own HIP streams, concurrent with the detector step on i)")
    ap.add_argument("--batch", type=int, default=None,
                    help="clouds (detector workload, default 4) / sweep pairs (loop workload, default 2 = the reference's batch_size, "
                         "liso_config.yml:121) per GPU and step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iou3d", action="store_true", help="skip the iou3d_nms section of the line")
    ap.add_argument("--no-fp32-leg", action="store_true",
                    help="loop workload, default dtype: skip the short parity legs (`parity_leg` = f32x3, `fp32_exact_leg` of the line)")
    ap.add_argument("--no-legs", action="store_true",
                    help="loop workload: skip the bounded configs[1] / [2] / [4] legs (`slim_leg`, `detector_leg`, `stress_leg`: child processes)")
    ap.add_argument("--uniform-clouds", action="store_true",
                    help="loop workload: every sweep with exactly 120000 loss-cloud points (default: 16 pairs of 116k-124k points, mean 120k, "
                         "so that the bucket padding and the per-signature graph caches are exercised as on real sweeps)")
    ap.add_argument("--loader", action="store_true",
                    help="loop workload: the sweep pairs live in PINNED HOST memory and are uploaded (H2D on a copy stream, one step before "
                         "they enter the pipeline's announcement window) as a DataLoader-fed run would (liso_cli.py:362-380); default: all "
                         "pairs resident in HBM (the `value` of the line)")
    ap.add_argument("--graph", action="store_true",
                    help="slim workload: replay forward+loss+backward from a hipGraph (host-independent step time)")
    ap.add_argument("--eager", action="store_true",
                    help="loop / detector workloads: launch every kernel eagerly (default: SLIM inference and the detector's "
                         "backbone + head + loss fwd/bwd replay from hipGraphs; bit-identical results)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` (no launcher): start N ranks as child processes BEFORE anything touches the GPU in this
    process (never exec from a process that has initialised HIP) and exit with the worst of their codes."""
    import socket

    if os.environ.get("LISO_DIST_BACKEND", "nccl") == "nccl":
        import torch  # (counting devices does not initialise HIP in this process)

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); RCCL wants one device per rank "
                     "(LISO_DIST_BACKEND=gloo shares devices between ranks: test mode only)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    sys.exit(max(abs(c) for c in codes))


# ---------------------------------------------------------------------------------------------------------------------
# algorithmic bytes / flops per launch of the timed kernel families; conv families report flops at the
# call site (units = 2 * M * N * K of the implicit GEMM)
def pfn_algorithmic_bytes(batch, n_points, grid, out_bytes):
    """pillar path: read the points once (N*C*4 B) + write the dense canvas once (64*G^2*s B) + occupancy (G^2*4 B)"""
    return batch * (n_points * 4 * 4 + 64 * grid * grid * out_bytes + grid * grid * 4)


def slim_algorithmic_bytes(batch, n_points, grid, levels=4, radius=3, directions=2):
    hw = (grid // 8) ** 2
    w2 = (2 * radius + 1) ** 2
    lookup = directions * batch * (levels * hw * w2 * 4 * 4 + hw * levels * w2 * 4)
    outputs = 6 * directions * batch * (grid * grid * 8 * 4 + hw * 6 * 4)
    return {"corr_lookup_fwd": lookup, "corr_lookup_fwd_tiled": lookup, "corr_lookup_bwd": lookup, "knn_query": 2 * n_points * 12 + n_points * 8,
            "raft_outputs_fwd": outputs, "raft_outputs_bwd": outputs}


KERNELS = {  # timer name -> (kernel description, bound, unit of `units`)
    "conv_bf16_fwd": ("conv_igemm_kernel<bf16> forward (NHWC implicit GEMM on v_mfma_f32_32x32x16_bf16, fused BN-apply "
                      "prologue / bias+ReLU+BN-statistics epilogue)", "mfma", "flop"),
    "conv_bf16_dgrad": ("conv_igemm_kernel<bf16> data gradient", "mfma", "flop"),
    "conv_bf16_wgrad": ("conv_wgrad_rs3_kernel<bf16> (3x3 / stride 1: all taps per block, row-stationary fragments, loader + MFMA waves) "
                        "and conv_wgrad_kernel<bf16> (other geometries): weight gradient, split over pixels", "mfma", "flop"),
    "conv_f32x3_fwd": ("conv_igemm_kernel<bf16x3> forward (fp32 tensors, hi/lo bf16 split, 3 MFMAs per product)", "mfma", "flop"),
    "conv_f32x3_fwd_roles": ("conv_roles_kernel<bf16x3> forward (3x3 / stride 1: loader waves + MFMA waves on double-buffered LDS, persistent "
                             "blocks; fp32 tensors, hi/lo bf16 split, 3 MFMAs per product)", "mfma", "flop"),
    "conv_f32x3_dgrad_roles": ("conv_roles_kernel<bf16x3> data gradient (3x3 / stride 1, mirrored taps)", "mfma", "flop"),
    "conv_f32x3_fwd_direct": ("conv_1x1_kernel / conv_taps_kernel<bf16x3> forward (1x1 layers; windows on <= 8 channels: fragments straight from "
                              "global memory, no LDS staging)", "mfma", "flop"),
    "conv_f32x3_dgrad_direct": ("conv_1x1_kernel<bf16x3> data gradient (1x1 layers)", "mfma", "flop"),
    "conv_bf16_fwd_roles": ("conv_roles_kernel<bf16> forward (3x3 / stride 1: loader waves + MFMA waves on double-buffered LDS, persistent "
                            "blocks; fused BN-apply prologue / bias+ReLU+BN-statistics epilogue)", "mfma", "flop"),
    "conv_bf16_dgrad_roles": ("conv_roles_kernel<bf16> data gradient (3x3 / stride 1, mirrored taps)", "mfma", "flop"),
    "conv_f32x3_dgrad": ("conv_igemm_kernel<bf16x3> data gradient", "mfma", "flop"),
    "conv_f32x3_wgrad": ("conv_wgrad_rs3_kernel<bf16x3> / conv_wgrad_kernel<bf16x3> weight gradient", "mfma", "flop"),
    "conv_f32_fwd": ("conv_igemm_kernel<f32> forward (exact fp32 on v_mfma_f32_32x32x2_f32)", "mfma", "flop"),
    "conv_f32_dgrad": ("conv_igemm_kernel<f32> data gradient (exact fp32)", "mfma", "flop"),
    "conv_f32_wgrad": ("conv_wgrad_kernel<f32> weight gradient (exact fp32)", "mfma", "flop"),
    "pfn_forward_scatter": ("pfn_forward_kernel (Linear+BN+ReLU+max + dense scatter from CSR feature rows)", "hbm", "bytes"),
    "pfn_decorate": ("pfn_decorate_kernel (+3 scan kernels)", "hbm", "bytes"),
    "corr_lookup_fwd": ("corr_lookup_fwd_kernel (on-the-fly 4-level correlation + bilinear lookup)", "hbm", "bytes"),
    "corr_lookup_fwd_tiled": ("corr_lookup_tiled_kernel (4 x 8 queries share the rows they correlate with: bf16x3 MFMA + bilinear lookup from LDS)", "hbm", "bytes"),
    "corr_lookup_bwd": ("corr_lookup_bwd_kernel (adjoint of the lookup into the dense volume gradient)", "hbm", "bytes"),
    "knn_query": ("knn_query_kernel (exact 1-NN, two-level bucket grid with z bins, 16 lanes per query)", "hbm", "bytes"),
    "raft_outputs_fwd": ("upsample_fwd_kernel (x8 bilinear upsampling + flow convention + concat of all RAFT iterations)", "hbm", "bytes"),
    "raft_outputs_bwd": ("upsample_bwd_x/y kernels (adjoint of the output assembly)", "hbm", "bytes"),
    "bev_dynamic_flow": ("bev_scatter_kernel + bev_mean_kernel (non-rigid flow, fixed-point scatter-mean)", "hbm", "bytes"),
    "dbscan_components": ("dbscan core/union/flatten kernels (grid-window DBSCAN)", "hbm", "bytes"),
    "dbscan_labels": ("dbscan_label_kernel", "hbm", "bytes"),
    "region_props": ("region_moments_kernel + region_props_kernel", "hbm", "bytes"),
    "kabsch_trafos": ("kabsch_moments_kernel + kabsch_solve_kernel (soft masks + weighted Kabsch)", "hbm", "bytes"),
    "fit_box_z": ("fit_z_kernel (points-in-box z extent)", "hbm", "bytes"),
    "conv_sparse_stem": ("cells_* + stem_taps_kernel + stem_gather_kernel (stride-2 convolution on the pillar canvas, occupied cells only; "
                         "bytes = the dense output written once + the occupancy map)", "hbm", "bytes"),
    "conv_sparse_dgrad": ("stem_dgrad_kernel (data gradient of that convolution at the occupied cells; bytes = the zero-filled canvas gradient)",
                          "hbm", "bytes"),
    "bn_fwd": ("bn_stats/finalize/apply kernels (BatchNorm2d+ReLU forward)", "hbm", "bytes"),
    "bn_bwd": ("bn_bwd_reduce/finalize/dx kernels (BatchNorm2d+ReLU backward)", "hbm", "bytes"),
}


def static_algorithmic_bytes(workload, batch, dtype_bytes):
    """bytes per launch for the families whose call sites do not report units (B = 1 in the loop)"""
    a = dict(slim_algorithmic_bytes(batch, N_POINTS, GRID, directions=1 if workload == "loop" else 2))
    a["pfn_forward_scatter"] = pfn_algorithmic_bytes(batch, N_POINTS, GRID, dtype_bytes)
    a["pfn_decorate"] = batch * N_POINTS * (16 + 48)
    a["bev_dynamic_flow"] = batch * (N_POINTS * (12 + 12 + 8 + 1) + GRID * GRID * 16)
    a["dbscan_components"] = batch * GRID * GRID * (1 + 12 + 1 + 4 + 4)
    a["dbscan_labels"] = batch * GRID * GRID * (1 + 12 + 1 + 4 + 4 + 4)
    a["region_props"] = batch * GRID * GRID * 4
    a["kabsch_trafos"] = batch * N_POINTS * (12 + 8 + 1)
    a["fit_box_z"] = batch * N_POINTS * 12
    return a


def pmc_traffic(workload, patterns):
    """HBM bytes per launch of the roofline kernel(s) from the committed rocprofv3 PMC passes of this bench command
    (profiles/r0N_<workload>_pmc_{FETCH,WRITE}_SIZE.csv, newest round first; counters cannot be collected from inside the
    timed process).  MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
    wide streaming reads, so reads are doubled."""
    import csv
    import glob

    for rnd in sorted({os.path.basename(p)[:3] for p in glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc_*.csv"))},
                      reverse=True):
        total, ok = 0.0, True
        for pat in patterns:
            for kind, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
                path = os.path.join(ROOT, "profiles", f"{rnd}_{workload}_pmc_{kind}.csv")
                # (a pattern "a|b" pools the launches of several kernels of ONE family into one weighted mean; separate list entries
                # are kernels that run once each per unit and are summed)
                rows = [r for r in csv.DictReader(open(path)) if any(q in r["Kernel_Name"] for q in pat.split("|"))] if os.path.exists(path) else []
                if not rows:
                    ok = False
                    break
                # (round 2 on: one row per kernel instantiation = mean over its launches, weighted by the launch count)
                wts = [float(r.get("Launches") or 1.0) for r in rows]
                total += factor * 1024.0 * sum(float(r["Counter_Value"]) * w for r, w in zip(rows, wts)) / sum(wts)
            if not ok:
                break
        if ok:
            return total, f"profiles/{rnd}_{workload}_pmc_*.csv (separate rocprofv3 --pmc passes of this command; 2 x FETCH_SIZE + WRITE_SIZE, KiB)"
    return None, None


# (template argument 1 of the convolution kernels = arithmetic: 0 bf16, 1 f32x3; forward and data gradient share the kernel)
PMC_PATTERNS = {"conv_f32x3_fwd_roles": ["conv_roles_kernel<1,"], "conv_f32x3_dgrad_roles": ["conv_roles_kernel<1,"],
                "conv_f32x3_fwd_direct": ["conv_1x1_kernel", "conv_taps_kernel"], "conv_f32x3_dgrad_direct": ["conv_1x1_kernel"],
                "conv_bf16_fwd_roles": ["conv_roles_kernel<0,"], "conv_bf16_dgrad_roles": ["conv_roles_kernel<0,"],
                "conv_bf16_fwd": ["conv_igemm_kernel<0,"], "conv_bf16_dgrad": ["conv_igemm_kernel<0,"], "conv_bf16_wgrad": ["conv_wgrad_kernel<0,|conv_wgrad_rs3_kernel<0,"],
                "conv_f32x3_fwd": ["conv_igemm_kernel<1,"], "conv_f32x3_dgrad": ["conv_igemm_kernel<1,"], "conv_f32x3_wgrad": ["conv_wgrad_kernel<1,|conv_wgrad_rs3_kernel<1,"],
                "conv_f32_fwd": ["conv_igemm_kernel<2,"], "conv_f32_dgrad": ["conv_igemm_kernel<2,"], "conv_f32_wgrad": ["conv_wgrad_kernel<2,"],
                "knn_query": ["knn_query_kernel"], "corr_lookup_fwd": ["corr_lookup_fwd_kernel"], "corr_lookup_fwd_tiled": ["corr_lookup_tiled_kernel"],
                "pfn_forward_scatter": ["pfn_forward_kernel"],
                "dbscan_components": ["dbscan_core_kernel", "dbscan_union_tiled_kernel", "dbscan_flatten_kernel"]}


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_detector(trainer, pcls, targets, torch):
    from oracle.train_step import timed_detector_step

    sd = {k: v.detach().float().cpu() for k, v in trainer.net.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t1 = {k: v[:1].cpu() for k, v in targets.items()}
    secs, _ = timed_detector_step(sd, [pcls[0].cpu()], t1, GRID, BEV_RANGE)
    return {"value": 1.0 / secs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 frame ({N_POINTS} pts, {GRID}x{GRID} BEV) fwd+bwd, fp32, torch-CPU oracle, {secs:.2f} s"}


def cpu_baseline_slim(cfg, trainer, s0, s1, torch):
    from oracle.slim_step import timed_slim_step

    sd = {k: v.detach().cpu() for k, v in trainer.net.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    secs, _ = timed_slim_step(cfg, sd, s0, s1)
    return {"value": 2.0 / secs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 pair = 2 frames ({N_POINTS} pts each, {GRID}x{GRID} BEV), 1 SLIM train step fwd+bwd+RMSprop, "
                      f"fp32, torch-CPU port with explicit correlation volume + cKDTree, {secs:.2f} s"}


def cpu_baseline_loop(cfg, trainer, s0, s1, torch):
    """oracle/liso_loop.py: the same fused iteration on the host cores, ONE sweep pair (bounded sample)"""
    from oracle.liso_loop import timed_loop_step

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    slim_sd = {k: v.detach().float().cpu() for k, v in trainer.slim.state_dict().items()}
    runs = [timed_loop_step(cfg, slim_sd, trainer.detector.net.state_dict(), s0, s1, GRID, BEV_RANGE) for _ in range(3)]  # (~2 s each)
    runs.sort(key=lambda r_: r_[0])
    secs, stages, n = runs[1]  # the median of three samples (one sample moved 1.0-1.3 frames/s between runs of round 5)
    return {"value": 2.0 / secs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 sweep pair = 2 frames ({N_POINTS} pts each, {GRID}x{GRID} BEV), one fused LISO iteration (SLIM fwd, "
                      f"sklearn DBSCAN + moments + Kabsch, NMS, targets, detector fwd+bwd), fp32 torch-CPU port, median of 3 runs "
                      f"({', '.join(f'{r_[0]:.2f}' for r_ in runs)} s), {n} mined boxes",
            "stage_seconds": {k: round(v, 4) for k, v in stages.items()}}


def bench_iou3d(dev, torch, with_cpu=True):
    """BASELINE.json metric, second half (SURVEY.md 8d ii): boxes/s of one nms_gpu-equivalent call INCLUDING the device
    greedy sweep, pairs/s of the rotated IoU matrix, at N in {256, 1000, 4096}; random car-like boxes (the oracle's
    generator, xy in +-50 m), sorted by score; HIP events on the launch stream, inputs resident in HBM.  VALU fraction at
    SURVEY's fixed 600 flop per pair vs the 157.3 TFLOP/s fp32 vector peak; the C oracle (single thread, like the
    reference's serial boxes_iou_bev_cpu + greedy loop) timed beside it."""
    import numpy as np

    from liso_amd import iou3d_nms_cuda as M
    from oracle import iou3d as O

    out = {"flop_per_pair": 600, "valu_peak_tflops": VALU_PEAK_F32_TF, "thresh": 0.1, "sizes": {}}
    for n in (256, 1000, 4096):
        b, s = O.random_boxes(n, 1, 50.0)
        b = b[np.argsort(-s, kind="stable")]
        tb = torch.from_numpy(b).to(dev)
        iou = torch.zeros(n, n, device=dev)
        reps = 50 if n <= 1000 else 20

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, z in e:
                a.record()
                fn()
                z.record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(z) for a, z in e)
            return t[len(t) // 2] * 1e-3

        t_nms = timed(lambda: M.nms_gpu_device(tb, 0.1))
        t_iou = timed(lambda: M.boxes_iou_bev_gpu(tb, tb, iou))
        keep = torch.zeros(n, dtype=torch.int64)
        t0 = time.perf_counter()
        for _ in range(5):
            M.nms_gpu(tb, keep, 0.1)
        t_api = (time.perf_counter() - t0) / 5
        row = {"nms_boxes_per_s": n / t_nms, "nms_ms": 1e3 * t_nms, "nms_host_api_ms": 1e3 * t_api,
               "nms_pairs_per_s": n * (n - 1) / 2 / t_nms,
               "iou_matrix_pairs_per_s": n * n / t_iou, "iou_matrix_ms": 1e3 * t_iou,
               "iou_matrix_valu_frac": 600.0 * n * n / t_iou / (VALU_PEAK_F32_TF * 1e12)}
        if with_cpu:
            t0 = time.perf_counter()
            ref_keep = O.nms(b, 0.1)
            t_cpu_nms = time.perf_counter() - t0
            m = min(n, 1000)  # bounded sample of the matrix for the serial CPU code
            t0 = time.perf_counter()
            O.boxes_iou_bev(b[:m], b[:m])
            t_cpu_iou = time.perf_counter() - t0
            num = M.nms_gpu(tb, keep, 0.1)
            row.update({"cpu_oracle_nms_boxes_per_s": n / t_cpu_nms, "cpu_oracle_iou_pairs_per_s": m * m / t_cpu_iou,
                        "cpu_cores": 1, "keep_identical_to_oracle": bool(np.array_equal(keep[:num].numpy(), ref_keep))})
            # the UNMODIFIED reference TU (iou3d_cpu.cpp compiled by oracle/Makefile into oracle/_ref) when its .so travelled with the
            # tree: the reference's own CPU path timed beside the restatement, same boxes, same bounded sample, one core
            t0 = time.perf_counter()
            ref_mat = O.ref_boxes_iou_bev(b[:m], b[:m])
            t_ref = time.perf_counter() - t0
            if ref_mat is not None:
                row.update({"cpu_reference_iou_pairs_per_s": m * m / t_ref, "cpu_reference_kind": "reference",
                            "cpu_reference_sample": f"boxes_iou_bev_cpu of the compiled reference TU on {m} x {m} boxes, {t_ref:.3f} s"})
        out["sizes"][str(n)] = row
    return out


def _sparse_overflow(dev):
    from liso_amd.utils import mfma_conv as MC

    return bool(MC.sparse_stem_overflowed(dev))


def child_leg(extra, steps=10, warmup=3, timeout_s=420):
    """one bounded run of another workload of this script as a CHILD process (started, never exec'ed, from this GPU process; the
    parent idles meanwhile) -> its JSON line trimmed to the numbers a reader of the default line needs"""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-iou3d", "--no-legs"] + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-300:]}
        j = json.loads(lines[-1])
    except Exception as e:  # a leg must never take the headline line down with it
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    rf = j.get("roofline", {})
    out = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"],
           "dtype": j["dtype"], "workload": j["config"]["workload"], "batch_per_gpu": j["config"]["batch_per_gpu"],
           "launch": j["config"]["launch"], "final_loss": j.get("final_loss"),
           "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")},
           "wall_s": round(time.perf_counter() - t0, 1), "command": "python bench.py " + " ".join(cmd[2:])}
    if "roofline_pillars" in j:
        out["roofline_pillars"] = {k: j["roofline_pillars"].get(k) for k in ("achieved", "peak", "unit", "frac", "avg_pair_ms", "batch")}
    return out


class PinnedLoader:
    """`--loader` ("separately with a pinned-memory loader"; the reference feeds its step from DataLoader workers,
    liso_cli.py:362-380): the ring of sweep pairs lives in PINNED HOST memory.  Every step uploads the pairs that enter the pipeline's
    announcement window at the NEXT step -- fresh device tensors, H2D copies on a copy stream -- and hands the upload's event to the step
    that first announces them (`step_batch(..., inputs_ready=event)`: every stream of the pipeline that reads them waits for it).
    A pair is collated into ONE pinned buffer (one H2D copy) and lands in a ring of device staging slots, one per ring entry, whose
    tensor views are made once (per-tensor uploads into fresh tensors cost the step's host thread 0.3-0.7 ms: 4.40-4.84 ms per step
    against 4.13-4.2 here); a slot is refilled only behind an event of the caller's stream recorded after the detector step
    of its previous occupant was enqueued (every reader of a pair precedes that step; the ring is longer than the pairs in flight).  Nothing a step consumes was on
    the device before its upload."""

    def __init__(self, pairs, dev, torch):
        self.torch, self.dev = torch, dev
        # every pair collated into ONE pinned byte buffer (what a DataLoader's collate_fn + pin_memory hands over): one H2D copy per pair,
        # the tensors of the samples are views of the device buffer (256-byte aligned)
        self.host, self.layout, self.bytes_per_pair = [], [], []
        for pair in pairs:
            leaves, off = [], 0
            self._map(pair, lambda t: leaves.append(t.detach().cpu().contiguous()) or t)
            offs = []
            for t in leaves:
                offs.append(off)
                off += (t.numel() * t.element_size() + 255) // 256 * 256
            buf = torch.empty(max(off, 256), dtype=torch.uint8).pin_memory()
            for t, o in zip(leaves, offs):
                n = t.numel() * t.element_size()
                if n:
                    buf[o:o + n].copy_(t.reshape(-1).view(torch.uint8))
            self.host.append(buf)
            self.layout.append((self._map(pair, lambda t: torch.empty(t.shape, dtype=t.dtype, device="meta")), offs))  # (shapes / dtypes only)
            self.bytes_per_pair.append(sum(t.numel() * t.element_size() for t in leaves))
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.window, self.pending, self.uploaded_bytes, self.uploads, self.slots = {}, None, 0, 0, {}

    @classmethod
    def _map(cls, obj, fn):
        import torch

        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, dict):
            return {k: cls._map(v, fn) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(cls._map(v, fn) for v in obj)
        return obj

    @classmethod
    def _leaf_bytes(cls, obj):
        tot = [0]
        cls._map(obj, lambda t: tot.__setitem__(0, tot[0] + t.numel() * t.element_size()) or t)
        return tot[0]

    def _upload(self, indices):
        """ring entries `indices` (absolute pair numbers) -> their device slots, filled on the copy stream behind an event of the caller's
        stream (the slot's previous occupant was consumed by work queued there)"""
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.copy_stream.wait_event(ready)
        got = {}
        for j in indices:
            k = j % len(self.host)
            if k not in self.slots:  # the slot's device buffer and the samples' views of it: made once, refilled by every upload
                dbuf = torch.empty(self.host[k].shape, dtype=torch.uint8, device=self.dev)
                template, offs = self.layout[k]
                it = iter(offs)

                def view(t, dbuf=dbuf, it=it):
                    o, n = next(it), t.numel() * t.element_size()
                    return dbuf[o:o + n].view(t.dtype).view(t.shape)

                self.slots[k] = (dbuf, self._map(template, view))
            dbuf, views = self.slots[k]
            with torch.cuda.stream(self.copy_stream):
                dbuf.copy_(self.host[k], non_blocking=True)
            got[j] = views
            self.uploaded_bytes += self.bytes_per_pair[k]
            self.uploads += 1
        done = torch.cuda.Event()
        done.record(self.copy_stream)
        return got, done

    def stepper(self, trainer, batch, n_up):
        torch = self.torch
        assert 2 * batch + n_up <= len(self.host), "ring shorter than the pairs in flight"

        def step():
            cur = torch.cuda.current_stream(self.dev)
            i = step.count * batch
            step.count += 1
            need = range(i, i + batch + n_up)
            ready = None  # the copy stream's event behind the newest upload that this step announces: every pipeline stream waits for it
            if self.pending is not None:  # uploaded during the previous step
                got, ready = self.pending
                self.window.update(got)
                self.pending = None
            missing = [j for j in need if j not in self.window]
            if missing:  # (first step only: nothing was announced before it)
                got, ready = self._upload(missing)
                self.window.update(got)
            self.pending = self._upload(range(i + batch + n_up, i + 2 * batch + n_up))  # enters the window at the next step
            loss = trainer.step_batch([self.window[j] for j in range(i, i + batch)],
                                      upcoming=tuple(self.window[j] for j in range(i + batch, i + batch + n_up)), inputs_ready=ready)
            for j in range(i, i + batch):
                del self.window[j]
            return loss

        step.count = 0
        return step


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with matching values")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    ranks_seen = 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LISO_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)  # (test mode: ranks share the GPUs that exist)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the HIP path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  LISO_DIST_BACKEND=gloo: the same bench with several ranks sharing ONE GPU (tests/test_gpu_multirank.py;
        # RCCL refuses two ranks on one device) -- never used for a reported number.
        backend = os.environ.get("LISO_DIST_BACKEND", "nccl")
        if backend == "nccl":
            assert torch.cuda.device_count() >= world, f"{world} ranks on {torch.cuda.device_count()} GPU(s): RCCL wants one device per rank"
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        # what the process group REALLY spans: every rank adds a one (a scaling record must show `ranks_seen` == n_gpus)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"process group of {ranks_seen} ranks, expected {world}"

    from liso_amd import _lib as L
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg


    if args.workload == "iou3d":
        if rank == 0:
            r = bench_iou3d(dev, torch, with_cpu=not args.no_cpu_baseline)
            n = "1000"
            print(json.dumps({"metric": "iou3d_nms boxes/s", "value": r["sizes"][n]["nms_boxes_per_s"], "unit": "boxes/s",
                              "n_gpus": 1, "steps": 50, "warmup": 3, "ms_per_step": r["sizes"][n]["nms_ms"],
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "iou3d_nms: rotated NMS incl. device greedy sweep, 1000 random boxes, IoU 0.1"},
                              "iou3d_nms": r}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    if args.workload == "stress":  # BASELINE configs[4]: 300k points, 1024^2 BEV (module constants: the byte formulas read them)
        global N_POINTS, GRID
        N_POINTS, GRID = 300000, 1024
    cfg = default_cfg(grid=GRID, bev_range_m=BEV_RANGE)
    torch.manual_seed(0)  # identical initial weights on every rank (DDP also broadcasts them)
    s0 = s1 = pcls = targets = loader = None
    slim_exact = False
    overlap = False
    if args.workload == "slim":
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import SlimTrainer

        # the reference trains SLIM in fp32 (no autocast in slim/experiment.py): fp32 tensors; --dtype fp32 = every convolution on the
        # native fp32 MFMA (`slim_exact_leg`), default = three bf16 MFMAs per product (F32X3)
        slim_exact = args.dtype == "fp32"
        args.dtype = "fp32"
        batch = 1
        cfg = apply_slim_simple_knn_training(cfg)
        trainer = SlimTrainer(cfg, dev, use_graph=args.graph, exact=slim_exact)
        s0, s1 = slim_pair(2 + rank, dev, n_points=N_POINTS, grid=GRID, bev_range_m=BEV_RANGE)
        step = lambda: trainer.step(s0, s1)  # noqa: E731
        frames_per_step = 2 * batch
    elif args.workload == "loop":
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import LisoLoopTrainer

        batch = args.batch or 2  # sweep pairs per detector step: the reference's default batch_size (liso_config.yml:121)
        dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        cfg = apply_slim_simple_knn_training(cfg)
        overlap = not (args.eager or args.no_overlap)
        # --dtype fp32 = every convolution (SLIM and detector) on the native fp32 MFMA; --dtype f32x3 = fp32 tensors everywhere,
        # three bf16 MFMAs per product (the parity-conformant configuration with the highest throughput)
        trainer = LisoLoopTrainer(cfg, dev, compute_dtype=dtype, total_steps=2 * (args.steps + args.warmup) + 64, use_graph=not args.eager,
                                  overlap=overlap, infer_batch=max(1, args.lookahead - 1 - args.flow_ahead), flow_ahead=args.flow_ahead,
                                  exact={"fp32": True, "f32x3": False}.get(args.dtype))
        # a ring of different sweep pairs; step i trains on pair i while (overlap) pairs i+1 / i+2 are in the mining stages.  Like real
        # sweeps they differ in their point counts (116k-124k non-ground points, mean 120k): the loss clouds are bucket-padded
        # (8192 rows) before they reach the captured graphs, so the ring exercises two input signatures per graph cache
        n_up = batch * (2 + args.flow_ahead) + max(1, args.lookahead - 1 - args.flow_ahead) - 1  # pairs the pipeline looks at (step_batch)
        n_pairs = max(16, n_up + batch + 2)
        offs = [((i * 3203) % 4001) for i in range(n_pairs // 2)]  # 0 .. 4000, mirrored around the mean: mean exactly N_POINTS
        counts = [N_POINTS] * n_pairs if args.uniform_clouds else \
            [N_POINTS + (offs[i // 2] if i % 2 == 0 else -offs[i // 2]) for i in range(n_pairs // 2 * 2)] + [N_POINTS] * (n_pairs % 2)
        pairs = [slim_pair(2 + rank + 100 * i, dev, n_points=counts[i], grid=GRID, bev_range_m=BEV_RANGE) for i in range(n_pairs)]
        s0, s1 = pairs[0]
        counter = [0]

        def step():
            i = counter[0] * batch
            counter[0] += 1
            return trainer.step_batch([pairs[(i + k) % len(pairs)] for k in range(batch)],
                                      upcoming=tuple(pairs[(i + k) % len(pairs)] for k in range(batch, batch + n_up)))

        if args.loader:
            loader = PinnedLoader(pairs, dev, torch)
            step = loader.stepper(trainer, batch, n_up)  # noqa: E731
            pairs = pairs[:max(batch, trainer.infer_batch)]  # (the resident copies go; these few serve the eager event passes behind the timed region)
            torch.cuda.empty_cache()

        frames_per_step = 2 * batch
    else:
        from liso_amd.datasets.synthetic import detector_batch
        from liso_amd.trainer import DetectorTrainer

        stress = args.workload == "stress"
        if stress:
            cfg.data.num_point_channels = 5
        batch = args.batch or (2 if stress else BATCH_PER_GPU)
        dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        trainer = DetectorTrainer(cfg, dev, compute_dtype=dtype, total_steps=args.steps + args.warmup + 8, use_graph=not args.eager,
                                  exact=(args.dtype == "fp32"))
        pcls, targets = detector_batch(seed=1 + rank, batch=batch, device=dev, n_points=N_POINTS, grid=GRID,
                                       bev_range_m=BEV_RANGE)
        if stress:  # 5th channel: sweep time offset of a 10-sweep accumulation in [0, 0.5) s
            gen = torch.Generator().manual_seed(3 + rank)
            pcls = [torch.cat([p_, (torch.randint(0, 10, (p_.shape[0], 1), generator=gen).float() * 0.05).to(dev)], dim=1) for p_ in pcls]
        step = lambda: trainer.step(pcls, targets)  # noqa: E731
        frames_per_step = batch

    graph_note = None
    if args.workload == "slim" and args.graph:
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
            trainer = SlimTrainer(cfg, dev, use_graph=False, exact=slim_exact)
            step = lambda: trainer.step(s0, s1)  # noqa: E731
    main_stream = None
    if os.environ.get("LISO_MAIN_PRIORITY"):  # experiment: the detector step's stream above the SLIM inference stream
        torch.cuda.synchronize()
        main_stream = torch.cuda.Stream(device=dev, priority=int(os.environ["LISO_MAIN_PRIORITY"]))
        torch.cuda.set_stream(main_stream)
    precapture_steps = 0
    if args.workload == "loop" and not args.eager:
        # hipGraph captures are one-time setup, not steady state: the inference graph has one signature per (pairs per replay, largest
        # point-count bucket of the batch), the box-mining graph one per bucket, and the ring of 16 sweep pairs of 16 different sizes
        # meets them all within one round -- more steps than the driver's --warmup.  One round (+ the pipeline's depth) of untimed steps
        # in front of the W warm-up steps, the same number on every rank (the steps hold collectives), reported in
        # config.graph_captures; `inside_timed_region` shows that none was left for the timed steps.
        precapture_steps = n_pairs // batch + 2 + trainer.infer_batch
        for _ in range(precapture_steps):
            step()
    for _ in range(args.warmup):
        step()
    captures_after_warmup = None
    if args.workload == "loop" and not args.eager:  # (reported: a capture inside the timed region would be part of `value`)
        captures_after_warmup = (len(trainer._infer_graphs), trainer.mine_captures)

    graphed = (args.workload == "slim" and args.graph) or (args.workload in ("loop", "detector", "stress") and not args.eager)
    if not graphed:  # per-launch HIP events on the launch stream, inside the timed region
        L.TIMER.enable_all()
    L.TIMER.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # one event per step boundary on the caller's stream (no synchronisation, ~1 us each): the distribution of the step times next to
    # the mean that `value` is made of.  In the pipelined loop a boundary is where the detector step of that batch ends on its stream.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        loss = step()
        marks[k + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    step_times = {"median_ms": step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]),
                  "p90_ms": step_ms[min(len(step_ms) - 1, int(0.9 * len(step_ms)))], "min_ms": step_ms[0], "max_ms": step_ms[-1],
                  "mean_ms_events": sum(step_ms) / len(step_ms),
                  "timing": f"{args.steps} hipEvent pairs on the caller's stream, one per step (ms_per_step = wall clock over the same steps between "
                            "two synchronisations)"} if step_ms else None
    # data parallelism: what the gradient all-reduce COSTS a step, measured: the same steps once more with the collective skipped (the
    # replicas diverge from here on: the parameter checksums below were taken first), and the collective alone on an idle GPU
    dist_cost = None
    if world > 1:
        det = getattr(trainer, "detector", trainer)
        cs_early = torch.stack([p.detach().double().sum() for p in det.net.parameters()]).sum().reshape(1)
        if hasattr(det, "_reduce_gradients") and det.use_graph:
            det.skip_collective = True
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dist.barrier()
            t_nocomm = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(t_nocomm, op=dist.ReduceOp.MAX)
            det.skip_collective = False
            buf = det._flat_grad
            torch.cuda.synchronize()
            dist.barrier()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                dist.all_reduce(buf)
            b.record()
            torch.cuda.synchronize()
            dist_cost = {"ms_per_step_without_collective": 1e3 * float(t_nocomm.item()) / args.steps,
                         "allreduce_alone_ms": a.elapsed_time(b) / 10, "allreduce_bytes": buf.numel() * buf.element_size(),
                         "note": "exposed all-reduce time per step = ms_per_step - ms_per_step_without_collective (same steps, collective skipped, "
                                 "MAX over ranks); allreduce_alone_ms = the flat gradient buffer all-reduced 10 times on an otherwise idle GPU"}
    timed_in, event_steps = "the timed steps", args.steps
    if graphed:
        # the timed steps replay a hipGraph, inside which per-kernel events cannot be recorded: the same kernels are timed
        # over two eager forward+backward passes right after the timed region.  EVERY rank runs them and they issue no
        # collective and no optimizer update (update=False), so the ranks stay in lock step for the MAX reduction below.
        L.TIMER.enable_all()
        event_steps = 4
        for k in range(event_steps + 1):
            if k == 1:  # (the first eager pass after a run of graph replays is untimed: allocator growth, cold instruction caches)
                torch.cuda.synchronize()
                L.TIMER.reset()
            if args.workload == "slim":
                trainer.step(s0, s1, eager=True, update=False)
            elif args.workload == "loop":
                trainer.eager_pass_batch(pairs[:batch], also=tuple(pairs[batch:max(batch, trainer.infer_batch)]) if overlap else ())
            else:
                trainer.eager_pass(pcls, targets)
        torch.cuda.synchronize()
        timed_in = f"{event_steps} eager fwd+bwd passes after the timed region, one untimed pass first (the timed steps replay a hipGraph)"
    L.TIMER.disable_all()
    legs = {}
    if args.workload == "loop" and args.dtype == "bf16" and world == 1 and rank == 0 and not args.no_fp32_leg:
        # the same iteration in the two parity-conformant arithmetics (north_star: logits / flow within 1e-3 of the reference's fp32
        # path; certified at this size by tests/test_gpu_parity_full_size.py), bounded legs, each this script in a CHILD process:
        #   parity_leg      fp32 tensors everywhere, three bf16 MFMAs per product (F32X3), SLIM and detector  (--dtype f32x3)
        #   fp32_exact_leg  native fp32 MFMA (v_mfma_f32_32x32x2_f32), SLIM and detector                     (--dtype fp32)
        # (Until late in round 5 the legs ran IN this process, next to the headline's trainer: a second LisoLoopTrainer brings three more
        # streams, HIP maps streams onto 4 hardware queues, and two pipeline stages then take turns on one queue -- the leg measured
        # 7.0-7.7 ms per step where `python bench.py --dtype f32x3` measures 5.6.  A child process has the GPU's queues to itself.)
        torch.cuda.empty_cache()
        common = ["--lookahead", str(args.lookahead), "--flow-ahead", str(args.flow_ahead)] + (["--no-overlap"] if args.no_overlap else []) + \
                 (["--eager"] if args.eager else []) + (["--batch", str(args.batch)] if args.batch else [])
        wl = 2 + 16 // max(batch, 1)  # (once around the ring of 16 sweep pairs: every graph signature captured before the timed steps)
        for name, exact, label in (("parity_leg", False, "f32 via bf16x3 MFMA (fp32 tensors, hi*hi + hi*lo + lo*hi), SLIM and detector"),
                                   ("fp32_exact_leg", True, "f32 (exact: native fp32 MFMA v_mfma_f32_32x32x2_f32, SLIM and detector)")):
            leg = child_leg(["--dtype", "fp32" if exact else "f32x3"] + common,
                            steps=max(min(args.steps, 20), 1) if not exact else min(args.steps, 10), warmup=wl, timeout_s=600)
            if "error" not in leg:
                leg["dtype"] = label
                # (nothing in this leg measures parity: these are the tests that compare this arithmetic with the CPU oracle at this
                # size, bar 1e-3)
                leg["parity_tests"] = ["tests/test_gpu_parity_full_size.py::test_detector_logits_and_loss_at_full_size_match_fp64_oracle"
                                       f"[{'exact' if exact else 'x3'}]",
                                       "tests/test_gpu_parity_full_size.py::test_slim_last_iteration_flow_at_full_size_matches_cpu_oracle"
                                       f"[{'exact' if exact else 'x3'}]"]
                leg["note"] = "same launch structure and inputs as the headline line, in a child process"
            legs[name] = leg
    if args.workload == "loop" and args.dtype == "bf16" and world == 1 and rank == 0 and not args.no_legs:
        # BASELINE configs[1] / [2] / [4] as bounded legs: this script with --workload slim | detector | stress in a child process
        # (the SLIM training graph needs another runtime mode, see _graph_env), its JSON line trimmed to the numbers
        for name, extra in (("slim_leg", ["--workload", "slim", "--graph"]), ("slim_exact_leg", ["--workload", "slim", "--graph", "--dtype", "fp32"]),
                            ("detector_leg", ["--workload", "detector"]), ("stress_leg", ["--workload", "stress"])):
            legs[name] = child_leg(extra, steps=min(args.steps, 10))
        # SURVEY 8d's second measurement mode: the same loop fed from pinned host memory (H2D on a copy stream), child process
        legs["loader_leg"] = child_leg(["--loader"], steps=min(args.steps, 20), warmup=2 + 16 // max(batch, 1))
        if "error" not in legs["loader_leg"]:
            legs["loader_leg"]["note"] = ("the headline's iteration with the sweep pairs in PINNED HOST memory, uploaded one step before they enter "
                                          "the pipeline, one collated pinned buffer and one H2D copy per pair into a ring of device staging slots (bench.py "
                                          "PinnedLoader); `value` of the line itself has every pair resident in HBM")
    export_cost = None
    if args.workload == "loop" and world == 1 and rank == 0 and not args.no_legs:
        # what a flow EXPORT costs next to the loop's inference (the miner reads flow t0 -> t1 only; liso/slim/experiment.py:363-471
        # writes both directions): eager calls on one sweep pair incl. the pillar encoders, wall time per pair
        def wall(fn, reps=8):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / reps

        export_cost = {"forward_direction_point_flow_ms_per_pair": wall(lambda: trainer.slim.infer_point_flow_t0_t1(s0, s1)),
                       "both_directions_bev_maps_ms_per_pair": wall(lambda: trainer.slim.infer_export_predictions(s0, s1)),
                       "note": "eager launches, one pair, pillar encoders included (the loop replays the one-direction form from a "
                               "hipGraph, several pairs per replay); SLIM.infer_export_predictions feeds liso_amd.slim.flow_io.flow_export_dict"}
    checksums = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # data parallelism keeps the replicas identical: every rank's parameter checksum travels to rank 0 and into the line
        cs = cs_early  # (taken right behind the timed steps, before the collective-free steps of `dist_cost`)
        got = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(got, cs)
        checksums = [float(g.item()) for g in got]

    if rank == 0:
        durs = {k: L.TIMER.durations_ms(k) for k in list(L.TIMER.events) if L.TIMER.events[k]}
        totals = {k: sum(v) for k, v in durs.items()}
        per_unit = {k: L.TIMER.weighted_total_ms(k) for k in durs}  # launches that serve a batch of iterations count 1 / batch
        static_bytes = static_algorithmic_bytes(args.workload, 1 if args.workload == "loop" else batch, 2 if args.dtype == "bf16" else 4)
        key = max(per_unit, key=per_unit.get)  # the hand-written kernel family with the largest share of the step
        if args.workload in ("detector", "stress") and key.startswith("pfn"):
            key = "pfn_forward_scatter"
        kname, bound, unit = KERNELS.get(key, (key, "hbm", "bytes"))
        n_launch = max(len(durs[key]), 1)
        units = L.TIMER.units.get(key, [])
        if key == "knn_query" and units:  # queries per launch as counted at the call site
            nq = sum(units) / len(units)
            alg_total = n_launch * ((nq + N_POINTS) * 12 + nq * 8)
        elif units and len(units) == len(durs[key]):
            alg_total = float(sum(units))
        else:
            alg_total = float(static_bytes.get(key, 0)) * n_launch
        t_total = totals[key] * 1e-3
        if key == "pfn_forward_scatter" and "pfn_decorate" in totals:  # the pillar pass is two launches: one unit
            t_total += totals["pfn_decorate"] * 1e-3
        if bound == "mfma":
            achieved = alg_total / t_total / 1e12 if t_total > 0 else 0.0
            peak = MFMA_PEAK_BF16_TF / 3.0 if "f32x3" in key else VALU_PEAK_F32_TF if key.startswith("conv_f32_") else MFMA_PEAK_BF16_TF
            runit = "TFLOP/s"
        else:
            achieved = alg_total / t_total / 1e9 if t_total > 0 else 0.0
            peak, runit = HBM_PEAK_GBS, "GB/s"
        # the pipelined loop caps the persistent 3x3 convolutions of the captured SLIM inference at `infer_cus` compute units (default: half
        # the chip, liso_amd/trainer.py) and the event passes time them under the same cap: their roofline is the peak of the CUs they may
        # use.  (Only where the family holds nothing but those launches: the bf16 headline's F32X3 forward family; in --dtype f32x3 the
        # detector's uncapped launches share the family and the whole chip's peak stays.)
        cu_cap = None
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        if (args.workload == "loop" and args.dtype == "bf16" and key == "conv_f32x3_fwd_roles" and getattr(trainer, "infer_cus", 0)
                and bound == "mfma"):
            cu_cap = {"cus": trainer.infer_cus, "of": n_cu, "whole_chip_peak": peak, "whole_chip_frac": achieved / peak}
            peak = peak * trainer.infer_cus / n_cu
        traffic, traffic_src = pmc_traffic(args.workload, PMC_PATTERNS[key]) if key in PMC_PATTERNS else (None, None)
        workload = {
            "loop": ("fused LISO iteration (BASELINE configs[3]): SLIM fwd (no_grad; forward flow direction t0->t1 only, only the last "
                     "of the 6 RAFT iterations decoded -- what the box miner consumes) -> FlowClusterDetector (DBSCAN) -> NMS "
                     f"-> target maps (per sweep pair) -> ONE CenterPoint-pillar train step on the batch of {batch} sweep pairs per GPU "
                     "(the reference's batch_size), 120k-pt sweeps, 512x512 BEV"),
            "slim": ("SLIM scene-flow train step (BASELINE configs[1]): two 120k-pt KITTI-shaped clouds, 512x512 BEV "
                     "pillars, RAFT 6 iterations fwd+bw flow, kNN loss, fwd+bwd+RMSprop"),
            "detector": ("CenterPoint-pillar detector train step (BASELINE configs[2]): 120k-pt KITTI-shaped clouds, "
                         "512x512 BEV pillars, fwd+bwd+AdamW"),
            "stress": ("CenterPoint-pillar detector train step (BASELINE configs[4] on one GPU): nuScenes-shaped 300k-pt clouds "
                       "(x, y, z, intensity, time), 1024x1024 BEV pillars, max 40000 pillars per cloud (deterministic first-come "
                       "cap), fwd+bwd+AdamW"),
        }[args.workload]
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
            "dtype": ("bf16 (detector, BASELINE configs[2]) + f32 via bf16x3 MFMA (SLIM); north_star's 1e-3 on logits / flow is met by "
                      "`parity_leg` (f32 via bf16x3 MFMA everywhere) and `fp32_exact_leg`, not by the bf16 detector"
                      if args.workload == "loop" and args.dtype == "bf16" else
                      "f32 (exact: native fp32 MFMA v_mfma_f32_32x32x2_f32)" if args.dtype == "fp32" and (args.workload != "slim" or slim_exact) else
                      "f32 via bf16x3 MFMA" if args.workload == "slim" or args.dtype == "f32x3" else
                      "bf16 (BASELINE configs[4] names fp16: the same MFMA rate and storage width; the kernels take bf16 / fp32 tensors only)"
                      if args.workload == "stress" and args.dtype == "bf16" else args.dtype),
            "data": "synthetic",
            "config": {"workload": workload, "points_per_cloud": N_POINTS, "bev_grid": GRID, "batch_per_gpu": batch,
                       "frames_per_step_per_gpu": frames_per_step, "parallelism": f"dp{world}",
                       # what the process group really looked like (a multi-GPU record shows that RCCL saw `world` ranks)
                       "dist": {"world_size": dist.get_world_size() if world > 1 else 1, "ranks_seen": ranks_seen,
                                "devices_visible": torch.cuda.device_count(),
                                "backend": dist.get_backend() if world > 1 else None,
                                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and
                                os.environ.get("LISO_DIST_BACKEND", "nccl") == "nccl" else None,
                                "gradient_buckets": getattr(getattr(trainer, "detector", trainer), "n_grad_buckets", 1)},
                       **({"points_per_cloud_ring": [min(counts), max(counts)], "sweep_pairs_in_rotation": n_pairs,
                           "point_bucket_rows": trainer.infer_point_bucket, "inference_compute_units": getattr(trainer, "infer_cus", 0) or "all",
                           "graph_captures": {"inference": len(trainer._infer_graphs), "box_mining": trainer.mine_captures,
                                              "inside_timed_region": (len(trainer._infer_graphs) - captures_after_warmup[0]) +
                                              (trainer.mine_captures - captures_after_warmup[1]),
                                              "box_mining_eager_fallbacks": trainer.mine_eager_fallbacks,
                                              "precapture_steps": precapture_steps}}
                          if args.workload == "loop" else {}),
                       "launch": ("eager" if not graphed else "hipGraph replay of fwd+loss+bwd, eager RMSprop" if args.workload == "slim"
                                  else "hipGraph replays (SLIM inference; detector backbone+head+loss fwd/bwd), eager pillar encoder / "
                                       "flow clustering / AdamW" + (f"; 3-stage pipeline on 3 HIP streams: SLIM inference {max(1, args.lookahead - 1 - args.flow_ahead)} pairs per replay | "
                                                                    "clustering+NMS+targets 1-2 pairs ahead (fixed box slots, no host reads) | "
                                                                    "detector step on pair i" if args.workload == "loop" and overlap else "")),
                       "convolutions": "own MFMA implicit-GEMM kernels",
                       # the stride-2 layers that read a pillar canvas multiply occupied cells only; True here would mean a batch held
                       # more occupied cells than the cell lists' capacity and some were dropped (never, with the voxeliser's 40000 cap)
                       "sparse_canvas_convolutions": {"enabled": os.environ.get("LISO_SPARSE_STEM", "1") != "0",
                                                      "cell_capacity_exceeded": _sparse_overflow(dev)}},
            "final_loss": float(loss),
            "step_times": step_times,
            "roofline": {"kernel": kname, "bound": bound, "achieved": achieved, "peak": peak, "unit": runit,
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": 1e3 * t_total / n_launch,
                         "launches_per_step": sum(L.TIMER.weights.get(key, [1.0] * n_launch)) / max(event_steps, 1),
                         ("algorithmic_flop_per_launch" if bound == "mfma" else "algorithmic_bytes_per_launch"): alg_total / n_launch,
                         **({"algorithmic_bytes_per_launch": sum(L.TIMER.bytes[key]) / len(L.TIMER.bytes[key])}
                            if bound == "mfma" and L.TIMER.bytes.get(key) else {}),
                         # share of the SUMMED event-timed kernel time (the kernels are timed alone on one stream; the pipelined
                         # step overlaps three streams, so a quotient by the step time would not be a share)
                         "share_of_timed_kernels": per_unit[key] / max(sum(per_unit.values()), 1e-12),
                         "timed_in": timed_in,
                         "timed_kernels_ms_per_step": {k: round(v / max(event_steps, 1), 4) for k, v in
                                                       sorted(per_unit.items(), key=lambda kv: -kv[1])}},
        }
        # the WHOLE step against the matrix-core roofline: algorithmic flops of every convolution launch of a step (counted at the call
        # sites, 2 * M * N * K of the implicit GEMM; launches that serve several steps weighted by their share) / the time those flops
        # take at the peak of the arithmetic they run in, over the measured step time.  The HBM-bound stages add nothing to the
        # numerator: this is "how much of the step's time is accounted for by matrix math at peak".
        def _peak_tf(k):
            return MFMA_PEAK_BF16_TF / 3.0 if "f32x3" in k else VALU_PEAK_F32_TF if k.startswith("conv_f32_") else MFMA_PEAK_BF16_TF
        conv_keys = [k for k in durs if KERNELS.get(k, ("", "", ""))[1] == "mfma" and len(L.TIMER.units.get(k, [])) == len(durs[k])]
        flops_by = {k: sum(u * w for u, w in zip(L.TIMER.units[k], L.TIMER.weights.get(k, [1.0] * len(durs[k])))) / max(event_steps, 1)
                    for k in conv_keys}
        at_peak_ms = sum(1e3 * f / (_peak_tf(k) * 1e12) for k, f in flops_by.items())
        line["step_roofline"] = {"bound": "mfma", "algorithmic_flop_per_step": sum(flops_by.values()),
                                 "time_at_peak_ms": at_peak_ms, "ms_per_step": 1e3 * elapsed / args.steps,
                                 "frac": at_peak_ms / (1e3 * elapsed / args.steps),
                                 "peaks_tflops": {"bf16": MFMA_PEAK_BF16_TF, "f32 via bf16x3": MFMA_PEAK_BF16_TF / 3.0, "f32 exact": VALU_PEAK_F32_TF},
                                 "flop_by_family": {k: v for k, v in sorted(flops_by.items(), key=lambda kv: -kv[1])},
                                 "note": "sum over the step's convolution launches of flops / peak(arithmetic) divided by the measured step time"}
        if loader is not None:
            line["loader"] = {"pinned_host_pairs": len(loader.host), "uploads": loader.uploads, "h2d_bytes_total": loader.uploaded_bytes,
                              "h2d_bytes_per_step": loader.uploaded_bytes / max(loader.uploads, 1) * batch,
                              "copy": "torch non_blocking copies from pinned memory on a dedicated copy stream, issued one step ahead"}
            line["config"]["workload"] += " -- sweep pairs fed from PINNED HOST memory (--loader)"
        if dist_cost is not None:
            dist_cost["exposed_allreduce_ms_per_step"] = line["ms_per_step"] - dist_cost["ms_per_step_without_collective"]
            line["dist_cost"] = dist_cost
        if "pfn_forward_scatter" in totals:
            # the pillar path as its own HBM roofline: decorate (+ scans) and forward launches timed together over the event passes;
            # algorithmic bytes = points read once + dense canvas + occupancy written once
            n_p = max(len(durs["pfn_forward_scatter"]), 1)
            t_p = (totals["pfn_forward_scatter"] + totals.get("pfn_decorate", 0.0)) * 1e-3
            # bytes as counted at every launch (pcl_to_feature_grid.py): points read once + the dense canvas at ITS element size (fp32 for
            # the SLIM encoders' canvases, bf16 for the detector's in the default loop) + the occupancy map written once
            pu = L.TIMER.units.get("pfn_forward_scatter", [])
            bytes_total = float(sum(pu)) if len(pu) == len(durs["pfn_forward_scatter"]) else 0.0
            tr_p, src_p = pmc_traffic(args.workload, ["pfn_forward_kernel", "pfn_decorate_kernel"])
            line["roofline_pillars"] = {"kernel": "pfn_decorate_kernel (+3 scan kernels) + pfn_forward_kernel", "bound": "hbm",
                                        "achieved": bytes_total / t_p / 1e9 if t_p > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": (bytes_total / t_p / 1e9 / HBM_PEAK_GBS) if t_p > 0 else 0.0,
                                        "algorithmic_bytes_per_launch_pair": bytes_total / n_p, "avg_pair_ms": 1e3 * t_p / n_p,
                                        "launch_pairs": n_p, "traffic": tr_p, "traffic_source": src_p}
        if cu_cap is not None:
            line["roofline"]["compute_units"] = cu_cap
        if bound == "mfma" and "f32x3" in key:
            line["roofline"]["peak_note"] = ("fp32 tensors computed as 3 bf16 MFMAs per product (hi*hi + hi*lo + lo*hi): peak = dense "
                                             "bf16 MFMA / 3 in algorithmic fp32 flops (native f32 MFMA peak: 157.3 TFLOP/s)" +
                                             (f"; these launches are capped at {cu_cap['cus']} of {cu_cap['of']} compute units (the pipeline's partition, "
                                              "`compute_units`): peak scaled by that share" if cu_cap is not None else ""))
        if world == 1 and not args.no_iou3d:
            line["iou3d_nms"] = bench_iou3d(dev, torch, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "slim":
                line["cpu_baseline"] = cpu_baseline_slim(cfg, trainer, s0, s1, torch)
            elif args.workload == "detector":
                line["cpu_baseline"] = cpu_baseline_detector(trainer, pcls, targets, torch)
            elif args.workload == "stress":
                line["cpu_baseline"] = None  # (the 1024^2 oracle step takes minutes: the 512^2 `--workload detector` line carries the CPU port)
            else:
                line["cpu_baseline"] = cpu_baseline_loop(cfg, trainer, s0, s1, torch)
        if args.workload == "loop":
            line["mined_boxes_last_step"] = int(trainer.last_boxes.valid.sum())
        line.update(legs)
        if export_cost is not None:
            line["slim_flow_export_inference"] = export_cost
        if checksums is not None:
            line["replica_param_checksums"] = checksums
            line["replicas_identical"] = all(c == checksums[0] for c in checksums)
            line["dist_backend"] = os.environ.get("LISO_DIST_BACKEND", "nccl")
        if graph_note:
            line["config"]["launch"] = graph_note
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
