"""CPU, world_size 2, gloo: the data-parallel wiring of DetectorTrainer (sample sharding, one flat gradient bucket,
per-rank BatchNorm, identical replicas after the step).  The HIP pillar encoder cannot run on CPU, so this test (and
only this test) substitutes a deterministic torch stand-in for `pfn.forward`; everything downstream is the product."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StandInPfn(torch.nn.Module):
    def __init__(self, weight, gamma, beta):
        super().__init__()
        self.weight, self.gamma, self.beta = weight, gamma, beta

    def forward(self, pcl_t0, img_t0=None):
        outs = []
        for p in pcl_t0:  # any deterministic differentiable map cloud -> [64, 32, 32]
            g = torch.zeros(32 * 32, 4)
            idx = ((p[:, 0].clamp(-9.9, 9.9) + 10) / 20 * 32).long() * 32 + ((p[:, 1].clamp(-9.9, 9.9) + 10) / 20 * 32).long()
            g.index_add_(0, idx, p)
            outs.append((g @ self.weight[:, :4].T).T.reshape(64, 32, 32))
        x = torch.stack(outs) * self.gamma[None, :, None, None] + self.beta[None, :, None, None]
        return x, (x.abs().sum(1, keepdim=True) > 0).float()


def _make(seed):
    from liso_amd.trainer import DetectorTrainer
    from liso_amd.utils.config import default_cfg

    torch.manual_seed(seed)
    tr = DetectorTrainer(default_cfg(grid=32, bev_range_m=20.0), torch.device("cpu"), total_steps=8)
    lyr = tr.net.model.pfn.pts_voxel_encoder.pfn_layers[0]
    tr.net.model.pfn.forward = _StandInPfn(lyr.linear.weight, lyr.norm.weight, lyr.norm.bias).forward
    return tr


def _data(rank):
    from liso_amd.datasets.synthetic import detector_batch

    return detector_batch(10 + rank, 1, torch.device("cpu"), n_points=3000, grid=32, bev_range_m=20.0)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = _make(0)
    assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
    pcls, targets = _data(rank)
    for _ in range(2):
        tr.step(pcls, targets)
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    rm = tr.net.model.rpn.blocks[0][2].running_mean.clone()
    rms = [torch.zeros_like(rm) for _ in range(world)]
    dist.all_gather(rms, rm)
    if rank == 0:
        torch.save({"params": gathered, "rm": rms}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_gloo_step_keeps_replicas_identical_and_equals_hand_averaged_gradients(tmp_path, world):
    """world 4 = the rank count of `bench.py --gpus 4` (the driver's 1 / 2 / 4 / 8 scaling runs): sample sharding, SUM all-reduce,
    the 1 / world factor"""
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = torch.load(out)
    # replicas stay identical, BatchNorm buffers stay per-rank (the reference uses plain BatchNorm2d)
    assert all(torch.equal(r["params"][0], r["params"][k]) for k in range(1, world))
    assert all(not torch.equal(r["rm"][0], r["rm"][k]) for k in range(1, world))
    # and equal to one process that averages the ranks' gradients by hand
    n_threads = torch.get_num_threads()
    torch.set_num_threads(2)  # same oneDNN blocking as the workers
    try:
        _compare_with_hand_averaged(r, world)
    finally:
        torch.set_num_threads(n_threads)


def _compare_with_hand_averaged(r, world):
    tr = _make(0)
    data = [_data(k) for k in range(world)]

    for _ in range(2):
        tr.model.train()
        grads = []
        for d in data:  # (BatchNorm running statistics do not enter the training-mode forward: one process can play every rank)
            tr.optimizer.zero_grad(set_to_none=True)
            total, _, _ = tr.loss(*d)
            total.backward()
            grads.append([p.grad.clone() if p.grad is not None else None for p in tr.net.parameters()])
        for p, *gs in zip(tr.net.parameters(), *grads):
            if gs[0] is not None:
                p.grad = sum(gs) / float(world)
        tr.optimizer.step()
        tr.lr_scheduler.step()
    flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()])
    # BN running stats do not enter the training-mode forward, so hand-averaging reproduces DDP up to fp32 summation order
    # (AdamW's first steps move every weight by ~lr * g / |g|: where the averaged gradient is ~0, another summation order of the
    # ranks' gradients may flip the direction of a step of size lr.  The bulk must agree tightly, no entry by more than 2 steps.)
    diff = (flat - r["params"][0]).abs()
    lr = max(g["lr"] for g in tr.optimizer.param_groups)
    # Two ranks: g0 + g1 is the same number in either order -> tight everywhere.  Four ranks: the ring's order differs from the
    # sequential sum, and parameters whose true gradient is ZERO (every convolution bias in front of a BatchNorm) take +-lr steps in the
    # direction of that rounding noise: most entries tight, none further apart than the steps taken.
    close = diff <= 2e-6 + 2e-4 * r["params"][0].abs()
    frac = float(close.float().mean())
    assert (frac == 1.0 if world == 2 else frac > 0.8) and float(diff.max()) <= 2 * 2 * lr + 1e-6, (frac, float(diff.max()), lr)


# ---- SLIM trainer under DDP (gloo, 2 ranks).  The six HIP-backed ops are swapped for the CPU port of oracle/slim_step.py
# (test infrastructure); model, decoder, loss, DDP wiring, optimizer and the threshold all-reduce are the product.
def _slim_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    threads = torch.get_num_threads()
    torch.set_num_threads(2)
    try:
        from liso_amd.datasets.synthetic import slim_pair
        from liso_amd.trainer import SlimTrainer
        from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
        from oracle.slim_step import cpu_port

        cfg = apply_slim_simple_knn_training(default_cfg(grid=128, bev_range_m=40.0))
        cfg.SLIM.model.num_iters = 2
        torch.manual_seed(0)
        with cpu_port():
            tr = SlimTrainer(cfg, torch.device("cpu"))
            assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
            s0, s1 = slim_pair(30 + rank, torch.device("cpu"), n_points=4000, grid=128, bev_range_m=40.0)
            losses = [float(tr.step(s0, s1)) for _ in range(2)]
        flat = torch.cat([p.detach().flatten() for p in tr.net.parameters()])
        # the dynamicness-threshold histogram: every rank feeds its own points, all ranks must end with the update of
        # the global batch (movavg_cls_threshold.py:118-157 + the all-reduce of SURVEY.md 8e)
        from liso_amd.slim.slim_loss.movavg_cls_threshold import MovingAverageThreshold
        g = torch.Generator().manual_seed(100 + rank)
        thr = MovingAverageThreshold(num_train_samples=100, num_moving=1000)
        es, ed, sc = torch.rand(500, generator=g), torch.rand(500, generator=g), torch.rand(500, generator=g)
        thr.update(es, ed, None, sc, True, valid_mask=torch.rand(500, generator=g) > 0.2)
        hist = thr.moving_average_importance.clone()
        bias = thr.bias_counter.clone().reshape(1)
        # deferred form (hipGraph steps cannot hold a collective): record two updates, reduce + apply them behind the step -- must
        # equal two immediate updates on the same inputs
        now, later = MovingAverageThreshold(num_train_samples=100, num_moving=1000), MovingAverageThreshold(num_train_samples=100, num_moving=1000)
        later._defer = []
        for k in range(2):
            a, b_, c = torch.rand(300, generator=g), torch.rand(300, generator=g), torch.rand(300, generator=g)
            vm = torch.rand(300, generator=g) > 0.3
            now.update(a, b_, None, c, True, valid_mask=vm)
            later.update(a, b_, None, c, True, valid_mask=vm)
        assert float(later.bias_counter) == 0.0
        items, later._defer = later._defer, None
        later.apply_deferred(items)
        assert torch.equal(now.moving_average_importance, later.moving_average_importance) and torch.equal(now.bias_counter, later.bias_counter)
        g_flat = [torch.zeros_like(flat) for _ in range(world)]
        g_hist = [torch.zeros_like(hist) for _ in range(world)]
        g_bias = [torch.zeros_like(bias) for _ in range(world)]
        dist.all_gather(g_flat, flat), dist.all_gather(g_hist, hist), dist.all_gather(g_bias, bias)
        if rank == 0:
            torch.save({"params": g_flat, "hist": g_hist, "bias": g_bias, "losses": losses}, out)
    finally:
        torch.set_num_threads(threads)
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_slim_step(tmp_path):
    out = str(tmp_path / "ddp_slim.pt")
    mp.spawn(_slim_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert all(l == l for l in r["losses"])  # finite
    assert torch.equal(r["params"][0], r["params"][1])          # replicas identical after two steps
    assert torch.equal(r["hist"][0], r["hist"][1]) and float(r["hist"][0].abs().sum()) > 0.0  # global histogram update
    assert torch.equal(r["bias"][0], r["bias"][1])
    # ... and equal to ONE process seeing both ranks' points
    from liso_amd.slim.slim_loss.movavg_cls_threshold import MovingAverageThreshold
    parts = []
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        parts.append((torch.rand(500, generator=g), torch.rand(500, generator=g), torch.rand(500, generator=g),
                      torch.rand(500, generator=g) > 0.2))
    one = MovingAverageThreshold(num_train_samples=100, num_moving=1000)
    one.update(*[torch.cat([p[i] for p in parts]) for i in (0, 1)], None, torch.cat([p[2] for p in parts]), True,
               valid_mask=torch.cat([p[3] for p in parts]))
    assert torch.allclose(one.moving_average_importance, r["hist"][0], rtol=1e-5, atol=1e-9)
    assert torch.allclose(one.bias_counter, r["bias"][0][0])
