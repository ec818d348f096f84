"""The fused LISO iteration: SLIM -> flow clusters -> NMS -> targets -> detector step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_liso_loop_runs_and_trains_on_mined_boxes():
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.trainer import LisoLoopTrainer
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
    from liso_amd.utils.nms_iou import box_iou_matrix

    dev = torch.device("cuda")
    cfg = apply_slim_simple_knn_training(default_cfg(grid=256, bev_range_m=50.0))
    torch.manual_seed(0)
    tr = LisoLoopTrainer(cfg, dev, total_steps=8)
    s0, s1 = slim_pair(7, dev, n_points=40000, grid=256, bev_range_m=50.0)
    w0 = {k: v.clone() for k, v in tr.detector.net.state_dict().items()}
    losses = [float(tr.step(s0, s1)) for _ in range(2)]
    assert all(torch.isfinite(torch.tensor(losses)))
    boxes = tr.last_boxes
    assert boxes.shape[0] == 1
    # mined boxes are NMS-clean (pairwise BEV IoU <= 0.1) and at most 100 per sample
    b0 = boxes[0].drop_padding_boxes()
    assert b0.shape[0] <= 100
    if b0.shape[0] > 1:
        iou = box_iou_matrix(b0, b0)
        iou.fill_diagonal_(0.0)
        assert float(iou.max()) <= 0.1 + 1e-5
    changed = sum(float((v.float() - w0[k].float()).abs().sum()) for k, v in tr.detector.net.state_dict().items() if v.is_floating_point())
    assert changed > 0.0  # the detector stepped
    # the SLIM network is frozen in this loop
    assert all(p.grad is None for p in tr.slim.parameters())


def test_inference_flow_equals_the_training_forward():
    """SLIM.infer_point_flow_t0_t1 (one direction, last iteration only) returns the same per-point flow as the full
    forward's preds_fw[-1].aggregated_flow"""
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.slim.model.slim import SLIM
    from liso_amd.utils.config import default_cfg

    dev = torch.device("cuda")
    torch.manual_seed(1)
    net = SLIM(default_cfg(grid=256, bev_range_m=50.0), 100).to(dev).eval()
    s0, s1 = slim_pair(11, dev, n_points=30000, grid=256, bev_range_m=50.0)
    with torch.no_grad():
        full, _ = net(s0, s1, None)
        fast = net.infer_point_flow_t0_t1(s0, s1)
    a, b = full[-1].aggregated_flow, fast
    assert a.shape == b.shape == (1, 30000, 3)
    assert float((a - b).abs().max()) <= 1e-4 * max(float(a.abs().max()), 1e-6)


def test_padded_device_nms_keeps_the_same_boxes_as_the_reference_schedule():
    """perform_nms_on_shapes_padded (no host round trips) vs perform_nms_on_shapes (nms_iou.py:23-66): same survivors per
    sample, in the same order"""
    from liso_amd.kabsch.shape_utils import Shape
    from liso_amd.utils.nms_iou import perform_nms_on_shapes, perform_nms_on_shapes_padded

    g = torch.Generator().manual_seed(0)
    B, K = 3, 300
    pos = torch.cat([torch.rand(B, K, 2, generator=g) * 40 - 20, torch.zeros(B, K, 1)], -1)
    dims = torch.stack([torch.rand(B, K, generator=g) * 3 + 2, torch.rand(B, K, generator=g) + 1.5, torch.full((B, K), 1.5)], -1)
    rot = (torch.rand(B, K, 1, generator=g) * 2 - 1) * 3.14159
    probs = torch.rand(B, K, 1, generator=g)
    valid = torch.rand(B, K, generator=g) > 0.2
    valid[2] = False  # a sample without boxes
    boxes = Shape(pos=pos.cuda(), dims=dims.cuda(), rot=rot.cuda(), probs=probs.cuda(), valid=valid.cuda())
    ref = perform_nms_on_shapes(boxes.clone(), max_num_boxes=40, overlap_threshold=0.1, pre_nms_max_num_boxes=200)
    got = perform_nms_on_shapes_padded(boxes.clone(), max_num_boxes=40, overlap_threshold=0.1, pre_nms_max_num_boxes=200)
    assert got.valid.shape == (B, K)
    for b in range(B):
        r, o = ref[b].drop_padding_boxes(), got[b].drop_padding_boxes()
        assert r.shape == o.shape, (b, r.shape, o.shape)
        assert torch.equal(r.pos, o.pos) and torch.equal(r.probs, o.probs) and torch.equal(r.rot, o.rot)
    assert int(got.valid[2].sum()) == 0 and int(got.valid[0].sum()) > 5
