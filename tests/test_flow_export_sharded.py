"""SLIM inference / export sharded by sample over the ranks (liso/slim/experiment.py:330-332,351-353: `sample_idx % world_size !=
worker_id -> continue`, no collective): gloo, world size 2 and 3, on CPU.  The inference itself is the GPU path's business
(tests/test_gpu_flow_io.py); here a deterministic stand-in produces the prediction objects so that the SHARDING and the file format
are what is tested: the ranks' files are disjoint, together they are exactly the single-process export, byte-identical arrays."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_SAMPLES = 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loader():
    for i in range(N_SAMPLES):
        g = torch.Generator().manual_seed(100 + i)
        s0 = {"pcl_ta": {"pcl": torch.randn(1, 50, 4, generator=g)}}
        s1 = {"pcl_ta": {"pcl": torch.randn(1, 50, 4, generator=g)}}
        yield f"seq/{i:04d}", s0, s1  # (sub-folders as in the waymo export, experiment.py:466-468)


def _infer(s0, s1):
    def pred(a, b):
        flow = (a["pcl_ta"]["pcl"][:, :16, :2].reshape(1, 4, 4, 2) - b["pcl_ta"]["pcl"][:, :16, :2].reshape(1, 4, 4, 2))
        dyn = a["pcl_ta"]["pcl"][:, :16, 2].reshape(1, 4, 4)
        return [types.SimpleNamespace(modified_network_output=types.SimpleNamespace(static_flow=flow, dynamicness=dyn))]

    return pred(s0, s1), pred(s1, s0)


def _worker(rank, world, port, out_dir):
    from liso_amd.slim.flow_io import export_flow_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    written = export_flow_sharded(_infer, _loader(), out_dir, lambda: torch.tensor(0.25), np.array([100.0, 100.0]))  # (rank from the group)
    torch.save([str(p) for p in written], os.path.join(out_dir, f"written_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_export_shards_by_sample_index_without_communication(tmp_path, world):
    from liso_amd.slim.flow_io import export_flow_sharded, load_flow_npz, shard_of

    single = tmp_path / "single"
    files = export_flow_sharded(_infer, _loader(), single, torch.tensor(0.25), np.array([100.0, 100.0]), world_size=1, worker_id=0)
    assert len(files) == N_SAMPLES
    multi = tmp_path / "multi"
    multi.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(multi)), nprocs=world, join=True)
    per_rank = [torch.load(multi / f"written_{r}.pt") for r in range(world)]
    for r, names in enumerate(per_rank):  # rank r wrote exactly the samples with idx % world == r, in loader order
        assert [os.path.relpath(n, multi) for n in names] == [f"seq/{i:04d}.npz" for i in range(N_SAMPLES) if i % world == r]
        assert all(shard_of(i, world, r) == (i % world == r) for i in range(N_SAMPLES))
    assert sum(len(n) for n in per_rank) == N_SAMPLES
    for f in files:
        a, b = load_flow_npz(f), load_flow_npz(multi / os.path.relpath(f, single))
        assert sorted(a) == sorted(b) == sorted(["bev_raw_flow_t0_t1", "bev_raw_flow_t1_t0", "bev_dynamicness_t0_t1", "bev_dynamicness_t1_t0",
                                                 "static_threshold", "bev_range_m"])
        assert all(np.array_equal(a[k], b[k]) for k in a)
    # skip_existing (experiment.py:383-384): nothing is rewritten
    assert export_flow_sharded(_infer, _loader(), single, torch.tensor(0.25), np.array([100.0, 100.0]), world_size=1, worker_id=0,
                               skip_existing=True) == []
