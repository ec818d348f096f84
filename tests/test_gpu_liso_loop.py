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
