"""Empty and ragged inputs through the device ops that later rows added ("cover the edge cases the
reference tests -- empty and ragged inputs")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_pillar_encoder_ragged_batch_with_empty_and_out_of_range_clouds():
    from liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid import PointsPillarFeatureNetWrapper
    from liso_amd.utils.config import default_cfg
    from oracle import pillars as OP

    cfg = default_cfg(grid=64, bev_range_m=40.0)
    torch.manual_seed(0)
    net = PointsPillarFeatureNetWrapper(cfg).to(DEV).train()
    g = torch.Generator().manual_seed(1)
    clouds = [torch.rand(5000, 4, generator=g) * torch.tensor([40.0, 40, 4, 1]) - torch.tensor([20.0, 20, 2, 0]),
              torch.zeros(0, 4),                                               # an empty sweep
              torch.rand(300, 4, generator=g) + torch.tensor([500.0, 0, 0, 0]),  # entirely outside the BEV range
              torch.rand(37, 4, generator=g) * torch.tensor([2.0, 2, 1, 1])]
    bev, occ = net([c.to(DEV) for c in clouds])
    assert bev.shape == (4, 64, 64, 64) and torch.isfinite(bev).all()
    assert float(occ[1].sum()) == 0 and float(occ[2].sum()) == 0 and float(bev[1].abs().sum()) == 0
    lyr = net.pts_voxel_encoder.pfn_layers[0]
    ref, ref_occ, _ = OP.pillar_forward([c.numpy() for c in clouds], lyr.linear.weight.detach().cpu(), lyr.norm.weight.detach().cpu(),
                                        lyr.norm.bias.detach().cpu(), torch.zeros(64), torch.ones(64), True, (40.0, 40.0), (64, 64),
                                        cfg.data.z_pillar_cutoff_value)
    assert torch.equal(occ.cpu(), ref_occ)
    assert float((bev.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
    bev.square().sum().backward()  # backward with empty samples in the batch
    assert all(torch.isfinite(p.grad).all() for p in (lyr.linear.weight, lyr.norm.weight, lyr.norm.bias))
    # a batch that holds no point at all
    bev0, occ0 = net([torch.zeros(0, 4, device=DEV)])
    assert float(bev0.abs().sum()) == 0 and float(occ0.sum()) == 0


def test_knn_empty_reference_and_empty_query():
    from liso_amd.slim.slim_loss.knn_graph import KnnIndex

    q = torch.rand(100, 3, device=DEV)
    idx, d2 = KnnIndex(torch.zeros(0, 3, device=DEV), extent=[-1.0, -1.0, 1.0, 1.0]).query(q, return_dist_sqr=True)
    assert idx.shape == (100,) and bool((idx == 0).all()) and bool(torch.isnan(d2).all())
    idx, d2 = KnnIndex(torch.rand(50, 3, device=DEV)).query(torch.zeros(0, 3, device=DEV), return_dist_sqr=True)
    assert idx.shape == (0,) and d2.shape == (0,)


def test_bev_gather_and_weighted_moments_without_points():
    from liso_amd.slim.slim_loss.static_aggregation import BevGatherPlan, batched_grid_data_to_pointwise_data
    from liso_amd.slim.slim_loss.weighted_pc_alignment import _WeightedMoments, weighted_pc_alignment

    grid = torch.randn(2, 8, 8, 5, device=DEV, requires_grad=True)
    coors = torch.zeros(2, 0, 2, dtype=torch.int32, device=DEV)
    valid = torch.zeros(2, 0, dtype=torch.bool, device=DEV)
    out = batched_grid_data_to_pointwise_data(grid, coors, valid, 0.0, plan=BevGatherPlan(coors, valid, (8, 8)))
    assert out.shape == (2, 0, 5)
    (g,) = torch.autograd.grad(out.sum() + 0.0 * grid.sum(), grid)
    assert float(g.abs().sum()) == 0
    mom = _WeightedMoments.apply(torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV), torch.zeros(0, device=DEV))
    assert mom.shape == (16,) and float(mom.abs().sum()) == 0
    # all weights zero: the epsilon fall-back of the reference (:26-34) keeps the fit finite (identity-like transform)
    p = torch.rand(200, 3, device=DEV)
    T, nep = weighted_pc_alignment(p, p + 0.1, torch.zeros(200, device=DEV))
    assert bool(nep) and torch.isfinite(T).all()


def test_flow_cluster_detector_without_motion_returns_no_boxes():
    from liso_amd.datasets.synthetic import cluster_sample
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector
    from liso_amd.utils.config import default_cfg

    det = FlowClusterDetector(default_cfg(grid=512, bev_range_m=100.0)).cuda()
    sample, _ = cluster_sample(8, torch.device(DEV), batch=2, n_points=20000)
    # a static world seen from a static sensor: flow 0, odometry identity
    sample["gt"]["flow_ta_tb"] = torch.zeros_like(sample["gt"]["flow_ta_tb"])
    sample["gt"]["odom_ta_tb"] = torch.eye(4, dtype=torch.float64, device=DEV)[None].repeat(2, 1, 1)
    boxes = det(sample, global_step=1)
    assert boxes.valid.shape == (2, 0) and boxes.pos.shape == (2, 0, 3) and boxes.dims.shape == (2, 0, 3)
    single = {"pcl_ta": {k: v[0] for k, v in sample["pcl_ta"].items() if k != "pcl_is_valid"},
              "pcl_full_w_ground_ta": sample["pcl_full_w_ground_ta"][0],
              "gt": {k: v[0] for k, v in sample["gt"].items()}, "src_trgt_time_delta_s": sample["src_trgt_time_delta_s"][0]}
    b1 = det(single, global_step=1, is_batched=False)  # reference :105-112 un-batched call
    assert b1.valid.shape == (0,)


def test_slim_decoder_kernels_with_a_sample_of_padding_rows_only():
    """include/liso_slim_decode.h: a sample whose rows are all padding next to a small cloud (4 points, 3 padding rows) through the fused
    decoder and the static-points loss: zeros out and zero gradients for every padding row, gradients only at the points' pillars"""
    from liso_amd.slim.model.head_decoder import HeadDecoder
    from liso_amd.slim.model import fused_decode as FD
    from liso_amd.utils.config import default_cfg

    cfg = default_cfg(grid=32, bev_range_m=20.0)
    ext = np.array([-10.0, -10.0, 10.0, 10.0])
    dec = HeadDecoder(cfg.SLIM, "d", ext)
    S, N, G, K = 2, 7, 32, 4
    pc = torch.zeros(S, N, 4, device=DEV)
    pc[1, :K, :3] = torch.tensor([[1.0, -2.0, 0.3], [4.0, 3.0, -0.2], [-6.0, 1.5, 0.1], [2.5, 7.0, 0.4]])
    valid = torch.zeros(S, N, dtype=torch.bool, device=DEV)
    valid[1, :K] = True                                           # sample 0: nothing valid
    coors = ((pc[..., :2] + 10.0) / 20.0 * G).to(torch.int32).clamp(0, G - 1)
    filled = torch.zeros(S, G, G, dtype=torch.bool, device=DEV)
    filled[1, coors[1, :K, 0].long(), coors[1, :K, 1].long()] = True
    net = torch.randn(S, G, G, 8, device=DEV, requires_grad=True)
    odom = torch.eye(4, device=DEV)[None].repeat(S, 1, 1)
    out = dec(net, 0.5, pc=pc, pointwise_voxel_coordinates=coors, pointwise_valid_mask=valid, filled_pillar_mask=filled, odom=odom,
              inv_odom=odom, summaries=None, pointwise_only=True)
    for k in ("aggregated_flow", "static_flow", "dynamic_flow", "class_probs", "staticness"):
        assert torch.isfinite(out[k]).all() and float(out[k][0].abs().sum()) == 0.0 and float(out[k][1, K:].abs().sum()) == 0.0, k
    assert torch.allclose(out.class_probs[1, :K].sum(dim=-1), torch.ones(K, device=DEV), atol=1e-6)
    # (sample 0 has no point at all: its transform is undefined, as in the reference; nothing reads it through a valid row)
    loss = FD.static_points_loss_mean(pc[1:], valid[1:], out.static_flow[1:], out.staticness[1:], out.static_aggr_trafo[1:])
    assert torch.isfinite(loss)
    (loss + out.aggregated_flow.sum()).backward()
    assert torch.isfinite(net.grad).all()
    touched = net.grad.abs().sum(dim=-1) > 0
    assert 1 <= int(touched.sum()) <= K and not bool(touched[0].any())   # only the points' pillars receive a gradient


def test_bike_rollout_and_instance_norm_degenerate_shapes():
    """liso_bike_rollout_*: an empty batch and the minimum track length; liso_in_relu_*: one sample, one pixel row"""
    from liso_amd.slim.model.fused_norm import in_act
    from liso_amd.tracker.track_smoothing import BatchedBikeModel, smooth_track_bike_model

    pos = torch.zeros(0, 6, 3, device=DEV)
    m = BatchedBikeModel(batched_observed_track_pos=pos, batched_vehicle_length=torch.zeros(0, device=DEV), time_between_frames_s=0.1,
                         max_yaw_rate=1.5, max_velocity=50.0)
    assert m.forward().shape == (0, 6, 5)
    short = torch.randn(2, 3, 3, device=DEV)  # fewer frames than the smoother needs: inputs come back unchanged (reference :596-605)
    p, y, d = smooth_track_bike_model(batched_observed_pos_m=short, batched_valid_mask=torch.ones(2, 3, dtype=torch.bool, device=DEV),
                                      batched_observed_yaw_angle_rad=torch.zeros(2, 3, 1, device=DEV),
                                      batched_vehicle_length_m=torch.full((2,), 4.0, device=DEV), time_between_frames_s=0.1)
    assert torch.equal(p, short) and d.shape == (2, 3)
    norm = torch.nn.InstanceNorm2d(8, eps=1e-3, affine=True).to(DEV)
    x = torch.randn(1, 8, 1, 5, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = in_act(x, norm)
    ref = torch.relu(norm(x.detach()))
    assert torch.allclose(y, ref, atol=1e-5)
    y.sum().backward()
    assert torch.isfinite(x.grad).all()
