"""SLIM flow export / ingest in the reference's on-disk format.

Export: liso/slim/experiment.py:363-471 writes one `np.savez_compressed` per sample with the LAST RAFT iteration's
`modified_network_output.static_flow` / `.dynamicness` of both directions (`bev_raw_flow_t0_t1`, `bev_raw_flow_t1_t0`,
`bev_dynamicness_t0_t1`, `bev_dynamicness_t1_t0`, optionally the t0_t2 / t1_t2 pairs), `static_threshold` and `bev_range_m`.
Ingest: liso/datasets/torch_dataset_commons.py:590-688 looks every point's pillar up in the BEV flow (after spreading valid
flow into zero-flow neighbour pillars) and hands `flow_ta_tb` [N,3] to the box miner.
Both directions are plain tensor ops here (they run wherever their inputs live, so the fused loop keeps everything in
HBM and a file written by either implementation is readable by the other)."""
from pathlib import Path

import numpy as np
import torch


def voxelize_pcl(pcl, grid_range_m, grid_size):
    """liso/datasets/nuscenes/analyse_boxes.py:6-26 -> (int32 voxel coordinates [N,3] by truncation, in-grid mask [N])"""
    if torch.is_tensor(pcl):
        rng = torch.as_tensor(grid_range_m, dtype=pcl.dtype, device=pcl.device)
        gs = torch.as_tensor(grid_size, device=pcl.device)
        coors = ((pcl[:, :3] + 0.5 * rng) / rng * gs).to(torch.int32)
        ok = ((coors >= 0) & (coors < gs)).all(dim=1)
        return coors, ok
    coors = ((pcl[:, :3] + 0.5 * grid_range_m) / grid_range_m * grid_size).astype(np.int32)
    ok = ((coors >= 0) & (coors < np.asarray(grid_size))).all(axis=1)
    return coors, ok


def expand_valid_bev_flow_to_zero_flow_neighbor_pillars(bev_flow):
    """torch_dataset_commons.py:670-688 -- pillars whose flow is exactly zero take the flow of a valid 4-neighbour
    (shifts -1/+1 along rows then columns, in that order, with wrap-around like np.roll; a pillar filled by an earlier
    shift counts as valid for the later ones, as with the reference's masked array)."""
    if not torch.is_tensor(bev_flow):
        return expand_valid_bev_flow_to_zero_flow_neighbor_pillars(torch.from_numpy(np.asarray(bev_flow))).numpy()
    flow = bev_flow.clone()
    invalid = (flow == 0.0).all(dim=-1)
    for shift in (-1, 1):
        for axis in (0, 1):
            shifted, shifted_invalid = torch.roll(flow, shift, dims=axis), torch.roll(invalid, shift, dims=axis)
            take = (~shifted_invalid) & invalid
            flow = torch.where(take[..., None], shifted, flow)
            invalid = invalid & ~take
    return torch.where(invalid[..., None], torch.zeros_like(flow), flow)


def point_flow_from_bev(pcl, bev_flow, flow_grid_range_m):
    """torch_dataset_commons.py:627-667 -- per-point 3-D flow [N,3] (z = 0) from a BEV flow map [H,W,2]; points outside
    the flow grid get the mean flow of the points inside."""
    tensor_in = torch.is_tensor(pcl)
    p = pcl if tensor_in else torch.from_numpy(np.asarray(pcl))
    f = bev_flow if torch.is_tensor(bev_flow) else torch.from_numpy(np.asarray(bev_flow))
    f = expand_valid_bev_flow_to_zero_flow_neighbor_pillars(f.to(p.device))
    rng = torch.cat([torch.as_tensor(flow_grid_range_m, dtype=p.dtype, device=p.device).reshape(-1)[:2],
                     torch.tensor([1000.0], dtype=p.dtype, device=p.device)])
    coors, ok = voxelize_pcl(p, rng, (f.shape[0], f.shape[1], 1))
    r, c = coors[:, 0].clamp(0, f.shape[0] - 1).long(), coors[:, 1].clamp(0, f.shape[1] - 1).long()
    inside = f[r, c].to(torch.float32)
    n_in = ok.sum().clamp(min=1)
    mean_in = torch.where(ok[:, None], inside, torch.zeros_like(inside)).sum(dim=0) / n_in
    flow2d = torch.where(ok[:, None], inside, mean_in[None].expand_as(inside))
    flow3d = torch.cat([flow2d, torch.zeros_like(flow2d[:, :1])], dim=-1)
    return flow3d if tensor_in else flow3d.numpy()


def flow_export_dict(preds_fw, preds_bw, static_threshold, bev_range_m, src="t0", trgt="t1"):
    """the arrays experiment.py:389-404,453-468 saves for one sample pair (batch dimension squeezed)"""
    out = {
        f"bev_raw_flow_{src}_{trgt}": preds_fw[-1].modified_network_output.static_flow,
        f"bev_raw_flow_{trgt}_{src}": preds_bw[-1].modified_network_output.static_flow,
        f"bev_dynamicness_{src}_{trgt}": preds_fw[-1].modified_network_output.dynamicness,
        f"bev_dynamicness_{trgt}_{src}": preds_bw[-1].modified_network_output.dynamicness,
        "static_threshold": torch.as_tensor(static_threshold),
    }
    out = {k: torch.squeeze(v, dim=0).detach().cpu().numpy() if v.dim() > 0 else v.detach().cpu().numpy() for k, v in out.items()}
    out["bev_range_m"] = np.asarray(bev_range_m)
    return out


def save_flow_npz(target_file, content):
    target_file = Path(target_file)
    target_file.parent.mkdir(exist_ok=True, parents=True)
    np.savez_compressed(target_file, **content)


def load_flow_npz(path):
    with np.load(path, allow_pickle=True) as z:
        return {k: z[k] for k in z.files}


def add_flow_to_sample(sample_content, pred_content, flow_source, src_key="ta", target_key="tb", file_src="t0", file_trgt="t1"):
    """torch_dataset_commons.py:590-667 for one pair: sample_content[flow_source][f"flow_{a}_{b}"] for both directions,
    from the exported maps `bev_raw_flow_{file_src}_{file_trgt}` / `..._{file_trgt}_{file_src}`"""
    rng = pred_content["bev_range_m"]
    sample_content[flow_source] = {}
    for (a, b), (fa, fb) in (((src_key, target_key), (file_src, file_trgt)), ((target_key, src_key), (file_trgt, file_src))):
        sample_content[flow_source][f"flow_{a}_{b}"] = point_flow_from_bev(
            sample_content[f"pcl_{a}"], pred_content[f"bev_raw_flow_{fa}_{fb}"], rng)
    return sample_content


def shard_of(sample_idx, world_size=None, worker_id=None):
    """does this rank export sample `sample_idx`?  liso/slim/experiment.py:330-332,351-353: `sample_idx % world_size == worker_id`,
    no communication between the ranks (every rank walks the whole loader and skips the others' samples).  world_size / worker_id
    default to the initialised torch.distributed process group (a single process exports everything)."""
    if world_size is None or worker_id is None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            world_size, worker_id = dist.get_world_size(), dist.get_rank()
        else:
            world_size, worker_id = 1, 0
    assert 0 <= worker_id < max(world_size, 1), (worker_id, world_size)
    return world_size <= 1 or sample_idx % world_size == worker_id


def export_flow_sharded(infer, loader, target_dir, static_threshold, bev_range_m, world_size=None, worker_id=None, skip_existing=False):
    """The inference / export pass of experiment.py:323-361 + :363-471 for the (t0, t1) pairs: every rank walks `loader` -- an iterable
    of (sample_id, sample_t0, sample_t1) -- and runs `infer(sample_t0, sample_t1) -> (preds_fw, preds_bw)` (e.g.
    `SLIM.infer_export_predictions`: both flow directions, last RAFT iteration) on ITS samples only, writing
    `target_dir / sample_id.npz` in the reference's format.  The path shards by sample: no collective on it, by construction the files
    of N ranks are the files of one rank.  `static_threshold`: a value or a callable evaluated per sample (the model's
    `moving_dynamicness_threshold.value`).  -> the files this rank wrote."""
    target_dir = Path(target_dir)
    written = []
    with torch.no_grad():
        for sample_idx, (sample_id, sample_t0, sample_t1) in enumerate(loader):
            if not shard_of(sample_idx, world_size, worker_id):
                continue
            target_file = (target_dir / str(sample_id)).with_suffix(".npz")
            if skip_existing and target_file.exists():
                continue
            preds_fw, preds_bw = infer(sample_t0, sample_t1)
            thr = static_threshold() if callable(static_threshold) else static_threshold
            save_flow_npz(target_file, flow_export_dict(preds_fw, preds_bw, thr, bev_range_m))
            written.append(target_file)
    return written
