"""Two ranks on ONE GPU (gloo carries the collectives, both processes compute on cuda:0): the data-parallel wiring of the
SLIM step that the RCCL runs of bench.py use with one GPU per rank -- the eager step under DistributedDataParallel
(bench.py's default) and the hipGraph step (`--graph`: parameter broadcast at start, flat gradient buffer all-reduced after
every replay)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, use_graph, static_aggr=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import SlimTrainer
        from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

        dev = torch.device("cuda:0")
        if static_aggr:  # the loss also updates the dynamicness threshold: one histogram all-reduce per update
            cfg = default_cfg(grid=128, bev_range_m=40.0)
            cfg.SLIM.model.use_static_aggr_flow_for_aggr_flow = True
            cfg.SLIM.losses.unsupervised.knn_on_static_penalty = 1.0  # (the threshold update compares static-aggregated and dynamic flow errors)
            cfg.SLIM.losses.unsupervised.artificial_labels.cross_entropy_penalty = 0.1
        else:
            cfg = apply_slim_simple_knn_training(default_cfg(grid=128, bev_range_m=40.0))
        torch.manual_seed(rank)  # different initial weights on purpose: the constructor must broadcast rank 0's
        tr = SlimTrainer(cfg, dev, use_graph=use_graph)
        assert (tr.model is tr.net) == use_graph  # no DDP wrapper in graph mode
        s0, s1 = slim_pair(50 + rank, dev, n_points=8000, grid=128, bev_range_m=40.0)
        for g in tr.optimizer.param_groups:  # leave the lr = 0 start of the warm-up schedule
            g["lr"] = 1e-3
        tr.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(tr.optimizer, lambda s: 1.0)
        losses = [float(tr.step(s0, s1)) for _ in range(3)]
        if use_graph:
            assert tr._graph is not None  # (the step really replayed a graph, also with the threshold updates of static aggregation)
        thr = tr.net.moving_dynamicness_threshold
        flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()] +
                         [thr.moving_average_importance.detach().flatten(), thr.bias_counter.detach().float().reshape(1)]).cpu()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            torch.save({"params": gathered, "losses": losses, "bias": float(thr.bias_counter)}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager_ddp", "hipgraph"])
def test_two_ranks_slim_step_keeps_replicas_identical(tmp_path, use_graph):
    out = str(tmp_path / "mr.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, use_graph), nprocs=2, join=True)
    r = torch.load(out)
    assert all(l == l for l in r["losses"])
    assert torch.equal(r["params"][0], r["params"][1])
    assert r["losses"][2] != r["losses"][0]  # the weights moved


@pytest.mark.timeout(900)
def test_two_ranks_slim_graph_step_with_static_aggregation_reduces_the_threshold_updates(tmp_path):
    """hipGraph step + several ranks + the dynamicness threshold learned from the loss: the captured step records the per-rank
    histogram increments, ONE all-reduce behind the replay makes them global and the updates are applied in order
    (movavg_cls_threshold.py: apply_deferred) -- parameters AND threshold buffers identical on both ranks, threshold updated"""
    out = str(tmp_path / "mr_sa.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, True, True), nprocs=2, join=True)
    r = torch.load(out)
    assert all(l == l for l in r["losses"])
    assert torch.equal(r["params"][0], r["params"][1])
    assert r["bias"] > 0.0  # the moving average really took updates


def _loop_worker(rank, world, port, out, use_graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import LisoLoopTrainer
        from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

        dev = torch.device("cuda:0")
        cfg = apply_slim_simple_knn_training(default_cfg(grid=256, bev_range_m=50.0))
        torch.manual_seed(0)  # (the frozen SLIM network is part of the checkpoint every rank loads: same seed)
        tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=20, use_graph=use_graph, overlap=bool(use_graph))
        if use_graph:  # two gradient buckets around a two-graph step unless LISO_GRAD_BUCKETS=1 asks for the single all-reduce
            assert tr.detector.n_grad_buckets == int(os.environ.get("LISO_GRAD_BUCKETS", "2"))
            assert tr.detector.optimizer.grad_scale == 0.5  # (the mean over ranks is folded into the AdamW launch)
        pairs = [slim_pair(60 + 10 * rank + i, dev, n_points=30000, grid=256, bev_range_m=50.0) for i in range(3)]
        losses = [float(tr.step(*pairs[i % 3], upcoming=(pairs[(i + 1) % 3], pairs[(i + 2) % 3]))) for i in range(5)]
        flat = torch.cat([p.detach().float().flatten() for p in tr.detector.net.parameters()]).cpu()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ls = [torch.zeros(5, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ls, torch.tensor(losses, dtype=torch.float64))
        if rank == 0:
            torch.save({"params": gathered, "losses": ls}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager_ddp", "graphs_pipeline_flat_allreduce"])
def test_two_ranks_liso_loop_keeps_detector_replicas_identical(tmp_path, use_graph):
    """the fused LISO iteration on two ranks with different sweeps: eager + DistributedDataParallel, and bench.py's default --
    hipGraph replays, the three-stream pipeline, ONE all-reduce of the flat gradient buffer, the one-launch AdamW"""
    out = str(tmp_path / "mr_loop.pt")
    mp.spawn(_loop_worker, args=(2, _free_port(), out, use_graph), nprocs=2, join=True)
    r = torch.load(out)
    assert all(bool(torch.isfinite(l).all()) for l in r["losses"])
    assert not torch.equal(r["losses"][0], r["losses"][1])  # the ranks saw different sweeps
    assert torch.equal(r["params"][0], r["params"][1])      # ... and took the same (averaged) steps
    if use_graph:
        # the bucketed schedule (two graphs, the large bucket reduced between them) changes WHEN the sums travel, not what they are:
        # same losses and parameters, bit for bit, as the single all-reduce behind one graph
        out1 = str(tmp_path / "mr_loop_1bucket.pt")
        os.environ["LISO_GRAD_BUCKETS"] = "1"
        try:
            mp.spawn(_loop_worker, args=(2, _free_port(), out1, use_graph), nprocs=2, join=True)
        finally:
            os.environ.pop("LISO_GRAD_BUCKETS", None)
        r1 = torch.load(out1)
        assert torch.equal(r["params"][0], r1["params"][0]) and all(torch.equal(a, b) for a, b in zip(r["losses"], r1["losses"]))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload,world", [("loop", 2), ("detector", 2), ("loop", 4)])
def test_bench_ranks_on_one_gpu_reports_world_and_identical_replicas(workload, world):
    """`python bench.py --gpus 2 | 4` through its own rank spawner, all ranks on cuda:0, gloo carrying the collectives
    (LISO_DIST_BACKEND=gloo: RCCL refuses two ranks on one device; the default backend "nccl" = RCCL is what the driver's
    multi-GPU runs use).  Exercises the world > 1 code of bench.py: barrier-bracketed timing, MAX over ranks, the flat-buffer
    gradient all-reduce after every replay, rank-0-only line.  The line must report n_gpus 2 and replicas that agree."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LISO_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "2", "--workload", workload,
                        "--no-cpu-baseline", "--no-iou3d"], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["config"]["parallelism"] == f"dp{world}" and line["scaling"] == "weak"
    assert line["dist_backend"] == "gloo" and len(line["replica_param_checksums"]) == world
    if workload == "loop":  # the bucketed schedule: two graphs, the large bucket's all-reduce between them (trainer.py)
        assert line["config"]["dist"]["gradient_buckets"] == 2 and line["config"]["dist"]["world_size"] == world
    assert line["replicas_identical"] is True
    assert line["value"] > 0 and line["final_loss"] == line["final_loss"]
    # round 6: the line itself shows how many ranks the process group spanned and what the gradient collective cost a step
    assert line["config"]["dist"]["ranks_seen"] == world and line["config"]["dist"]["devices_visible"] >= 1
    if workload != "slim":
        dc = line["dist_cost"]
        assert dc["allreduce_bytes"] > 1 << 20 and dc["allreduce_alone_ms"] > 0 and dc["ms_per_step_without_collective"] > 0
        assert abs(dc["exposed_allreduce_ms_per_step"] - (line["ms_per_step"] - dc["ms_per_step_without_collective"])) < 1e-9
    assert line["step_times"]["median_ms"] > 0 and line["step_roofline"]["algorithmic_flop_per_step"] > 0


@pytest.mark.timeout(600)
def test_bench_loader_run_trains_like_the_resident_run():
    """`bench.py --loader` (SURVEY 8d's second measurement mode): the sweep pairs live in pinned host memory, every pair is one collated
    buffer uploaded on a copy stream into a ring of device staging slots, and the step waits for the upload's event
    (`step_batch(..., inputs_ready=...)`).  Same iteration, same data: the loss after the same number of steps equals the resident
    run's, and the line reports the uploads."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = []
    for extra in ([], ["--loader"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-iou3d",
                            "--no-legs"] + extra, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    resident, fed = out
    assert "loader" not in resident and fed["loader"]["uploads"] > 0 and fed["loader"]["h2d_bytes_per_step"] > 1 << 20
    assert fed["final_loss"] == resident["final_loss"], (fed["final_loss"], resident["final_loss"])
    assert fed["mined_boxes_last_step"] == resident["mined_boxes_last_step"]


def test_rccl_process_group_next_to_graph_captures_and_replays():
    """RCCL itself (backend "nccl"), one rank on the one GPU there is: a live process group (communicator + watchdog thread) while the
    loop captures and replays its three hipGraphs, and an all-reduce of the detector's flat gradient buffer between replays -- the calls
    a second rank adds (two ranks cannot share a device under RCCL; the multi-rank logic itself runs over gloo above)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "rccl_single_rank_check.py")], cwd=root, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok: 10 loop steps" in r.stdout
