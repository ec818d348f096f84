"""Parity at BASELINE size (120k-point clouds, 512 x 512 BEV) against the CPU oracle, for BOTH fp32 arithmetic modes of the
convolutions -- "exact" (native fp32 MFMA) and "x3" (fp32 tensors, three bf16 MFMAs per product; what `bench.py --dtype f32x3`
and the default line's `parity_leg` run).  north_star's bar: logits and flow within 1e-3 (relative to the map's largest value).

  * detector: raw logit maps of all four heads + the loss, B = 1, train-mode BatchNorm, against oracle/train_step.py in fp64
    (rpn.py:137-146, center_head.py:109-117, centerpoint_loss.py:13-136);
  * SLIM: the network output of the last RAFT iteration (logits + static / dynamic flow, both directions) in training mode, and
    the per-point flow the box miner consumes, against oracle/slim_step.py::cpu_port (pinned to the reference's outputs by
    tests/test_oracle_slim.py) in fp32 (raft_mod.py:82-259, head_decoder.py:410-496).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HEADS = ("pos", "dims", "rot", "probs")
N_POINTS, GRID, RANGE = 120000, 512, 100.0


def _rel(a, b):
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


@pytest.mark.parametrize("fmode", ["exact", "x3"])
def test_detector_logits_and_loss_at_full_size_match_fp64_oracle(fmode):
    from liso_amd.datasets.synthetic import detector_batch
    from liso_amd.trainer import DetectorTrainer
    from liso_amd.utils import mfma_conv as MC
    from liso_amd.utils.config import default_cfg
    from oracle.train_step import detector_forward_loss, prepare_state

    dev = torch.device("cuda:0")
    prev = MC.fp32_mode()
    try:
        torch.manual_seed(31)
        tr = DetectorTrainer(default_cfg(grid=GRID, bev_range_m=RANGE), dev, compute_dtype=torch.float32, total_steps=8,
                             exact=(fmode == "exact"))
        pcls, targets = detector_batch(36, 1, dev, n_points=N_POINTS, grid=GRID, bev_range_m=RANGE)
        sd64 = prepare_state(tr.net.state_dict(), torch.float64)
        tr.model.train()
        with torch.no_grad():
            _, _, raw, _ = tr.net(None, pcls, None, decode=False)
        total, _, _ = tr.loss(pcls, targets)
    finally:
        MC.set_fp32_mode(prev)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        ref_total, raw64, _ = detector_forward_loss(sd64, [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}, GRID, RANGE,
                                                    training=True, dtype=torch.float64)
    errs = {h: _rel(raw[h], raw64[h]) for h in HEADS}
    print(f"[{fmode}] full-size detector logits vs fp64 oracle (max |diff| / max |ref|):", {k: f"{v:.2e}" for k, v in errs.items()},
          f"loss {abs(float(total) - float(ref_total)) / abs(float(ref_total)):.2e}")
    for h in HEADS:
        assert raw[h].shape == raw64[h].shape
        assert errs[h] <= 1e-3, (fmode, h, errs[h])
    assert abs(float(total) - float(ref_total)) <= 1e-3 * abs(float(ref_total))


def _to_cpu(s):
    return {k: (_to_cpu(v) if isinstance(v, dict) else [t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in s.items()}


@pytest.mark.parametrize("fmode", ["exact", "x3"])
def test_slim_last_iteration_flow_at_full_size_matches_cpu_oracle(fmode):
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.slim.model.slim import SLIM, get_network_input_pcls
    from liso_amd.utils import mfma_conv as MC
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
    from oracle.slim_step import cpu_port

    dev = torch.device("cuda:0")
    cfg = apply_slim_simple_knn_training(default_cfg(grid=GRID, bev_range_m=RANGE))
    torch.manual_seed(7)
    net = SLIM(cfg, num_train_samples=1000).to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    s0, s1 = slim_pair(12, dev, n_points=N_POINTS, grid=GRID, bev_range_m=RANGE)
    prev = MC.set_fp32_mode(fmode)
    try:
        with torch.no_grad():
            net.train()
            fw, bw, _ = net.raft_network(get_network_input_pcls(cfg, s0, "ta", to_device=dev), get_network_input_pcls(cfg, s1, "ta", to_device=dev))
            last_fw, last_bw = fw[-1].clone(), bw[-1].clone()
            net.eval()
            flow = net.infer_point_flow_t0_t1(s0, s1).clone()
    finally:
        MC.set_fp32_mode(prev)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with cpu_port(), torch.no_grad():
        ref = SLIM(cfg, num_train_samples=1000)
        ref.load_state_dict(sd)
        c0, c1 = _to_cpu(s0), _to_cpu(s1)
        ref.train()
        rfw, rbw, _ = ref.raft_network(get_network_input_pcls(cfg, c0, "ta"), get_network_input_pcls(cfg, c1, "ta"))
        ref.eval()
        rflow = ref.infer_point_flow_t0_t1(c0, c1)
    assert last_fw.shape == rfw[-1].shape == (1, GRID, GRID, 8)
    report = {}
    for tag, got, want in (("fw", last_fw, rfw[-1]), ("bw", last_bw, rbw[-1])):
        # channels of concat2network_output (head_decoder.py:37-65): 4 class logits | static flow (2) | dynamic flow (2)
        report[tag + "_logits"] = _rel(got[..., :4], want[..., :4])
        report[tag + "_flow"] = _rel(got[..., 4:], want[..., 4:])
    valid = s0["pcl_ta"]["pcl_is_valid"][0].cpu()
    report["point_flow"] = _rel(flow[0].cpu()[valid], rflow[0][valid])
    print(f"[{fmode}] full-size SLIM vs CPU oracle (max |diff| / max |ref|):", {k: f"{v:.2e}" for k, v in report.items()})
    for k, v in report.items():
        assert v <= 1e-3, (fmode, k, v)
