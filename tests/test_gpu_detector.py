"""GPU parity: the whole CenterPoint-pillar train step (HIP pillar path + BEV backbone + head + decode + loss) against
the CPU oracle train step on the same seeded inputs and the same initial state_dict."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(grid, rng, B, n, dtype=torch.float32, seed=0):
    from liso_amd.datasets.synthetic import detector_batch
    from liso_amd.trainer import DetectorTrainer
    from liso_amd.utils.config import default_cfg

    torch.manual_seed(seed)
    cfg = default_cfg(grid=grid, bev_range_m=rng)
    tr = DetectorTrainer(cfg, torch.device("cuda:0"), compute_dtype=dtype, total_steps=20)
    pcls, targets = detector_batch(seed + 5, B, torch.device("cuda:0"), n_points=n, grid=grid, bev_range_m=rng)
    return tr, pcls, targets


def _rel(a, b):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


def test_train_step_matches_cpu_oracle_fp32():
    """Forward (loss) within 1e-3 of the fp64 oracle.  Gradients: the loss is piecewise smooth -- 28 ReLU layers with
    ~10^7 activations, a handful of which sit within fp32 rounding of their kink; whether such an activation counts as
    positive decides if the (possibly large) upstream gradient of that pixel enters every parameter gradient upstream.
    Measured: the GPU's inputs to `center_head.shared_conv` and the gradient arriving at its output agree with the
    oracle to 3e-6 / 1e-6, the block's own backward agrees with an fp64 replay to 3e-7, yet its bias gradient differs by
    2.5e-2 because of flipped kinks (scripts/debug_head.py).  The tolerance is therefore the oracle's OWN sensitivity:
    the fp64 oracle is re-run with every parameter perturbed by the relative size of the measured forward mismatch
    (3e-6), and each GPU gradient must be within max(1e-3, 3x the largest change that perturbation causes)."""
    from oracle.train_step import detector_forward_loss, prepare_state

    tr, pcls, targets = _setup(128, 100.0, 2, 20000)
    sd0 = tr.net.state_dict()
    tr.model.train()
    total, losses, _ = tr.loss(pcls, targets)
    total.backward()
    cpu_pcls = [p.cpu() for p in pcls]
    t_cpu = {k: v.cpu() for k, v in targets.items()}
    sd64 = prepare_state(sd0, torch.float64)
    ref64, _, _ = detector_forward_loss(sd64, cpu_pcls, t_cpu, 128, 100.0, dtype=torch.float64)
    ref64.backward()
    assert abs(float(total.detach()) - float(ref64.detach())) <= 1e-3 * abs(float(ref64.detach()))
    names = dict(tr.net.named_parameters())
    keys = [k for k, p in names.items() if p.grad is not None and sd64[k].grad is not None
            and float(sd64[k].grad.abs().max()) >= 1e-6]  # conv biases in front of a BatchNorm: true gradient == 0
    noise = 0.0
    gen = torch.Generator().manual_seed(1)
    for _ in range(4):  # kink sensitivity of the reference arithmetic itself
        sdp = prepare_state(sd0, torch.float64)
        with torch.no_grad():
            for k in keys:
                sdp[k].mul_(1.0 + 3e-6 * torch.randn(sdp[k].shape, generator=gen, dtype=torch.float64))
        refp, _, _ = detector_forward_loss(sdp, cpu_pcls, t_cpu, 128, 100.0, dtype=torch.float64)
        refp.backward()
        noise = max(noise, max(_rel(sdp[k].grad, sd64[k].grad) for k in keys))
    worst = max(_rel(names[k].grad, sd64[k].grad) for k in keys)
    print(f"worst gradient error {worst:.2e}; oracle sensitivity to a 3e-6 perturbation {noise:.2e}")
    for k in keys:
        assert _rel(names[k].grad, sd64[k].grad) <= max(1e-3, 3 * noise), (k, _rel(names[k].grad, sd64[k].grad), noise)
    assert len(keys) > 60
    # the last layers see no kinks downstream except their own head: tight
    for k in ("model.center_head.tasks.0.rot.3.weight", "model.center_head.tasks.0.probs.3.bias"):
        assert _rel(names[k].grad, sd64[k].grad) < 1e-3, k


def test_eval_mode_gradients_tight():
    """With BatchNorm in eval mode (affine maps) the backward is well conditioned: every gradient within 1e-3."""
    from oracle.train_step import detector_forward_loss, prepare_state

    tr, pcls, targets = _setup(128, 100.0, 2, 20000, seed=2)
    with torch.no_grad():  # non-trivial running stats
        for m in tr.net.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.6, 1.4)
    tr.net.eval()
    total, _, _ = tr.loss(pcls, targets)
    total.backward()
    sd64 = prepare_state(tr.net.state_dict(), torch.float64)
    ref, _, _ = detector_forward_loss(sd64, [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}, 128, 100.0,
                                      training=False, dtype=torch.float64)
    ref.backward()
    assert abs(float(total) - float(ref)) <= 1e-3 * abs(float(ref))
    n = 0
    for k, p in tr.net.named_parameters():
        if p.grad is not None and sd64[k].grad is not None:
            assert _rel(p.grad, sd64[k].grad) < 1e-3, (k, _rel(p.grad, sd64[k].grad))
            n += 1
    assert n > 80


def test_bf16_step_runs_and_tracks_fp32():
    tr32, pcls, targets = _setup(256, 100.0, 2, 60000, torch.float32, seed=3)
    tr16, _, _ = _setup(256, 100.0, 2, 60000, torch.bfloat16, seed=3)
    tr16.net.load_state_dict(tr32.net.state_dict())
    tr32.model.train(), tr16.model.train()
    l32, _, _ = tr32.loss(pcls, targets)
    l16, _, _ = tr16.loss(pcls, targets)
    # bf16 BEV tensors over ~28 conv layers: a few 1e-2 relative on the loss is the documented budget
    assert abs(float(l16) - float(l32)) <= 5e-2 * abs(float(l32))
    a = [float(tr16.step(pcls, targets)) for _ in range(6)]
    assert all(np.isfinite(a)) and a[-1] < a[0]  # it trains


def test_bf16_detector_logit_error_vs_exact_fp32_at_full_size():
    """The headline configuration runs the detector with bf16 BEV tensors.  Its error is quantified here against the EXACT fp32
    path (native fp32 MFMA) on the bench's input -- 120k-point clouds, 512^2 BEV, train-mode BatchNorm, same weights: maximum and
    root-mean-square error of every head's raw logits relative to the head's largest / rms logit, and the loss.
    Measured (MI355X, random-init weights, B = 1): worst element 3.2e-2 ... 5.8e-2 of the head's range, rms error 4.4e-2 ... 5.4e-2
    of the head's rms logit, loss 8.7e-5 -- bf16 carries 8 significant bits (2^-9 = 2e-3 per rounding of every stored activation and
    every weight) through 17 convolution + train-mode BatchNorm layers whose outputs at initialisation are small differences of large
    sums.  Budget asserted here and documented in DESIGN.md section 5: worst element <= 1e-1, rms <= 8e-2, loss <= 5e-2.  north_star's
    1e-3 on logits is the requirement on the fp32 parity configuration (asserted in test_logits_match_oracle_* / the fixture tests)."""
    from liso_amd.utils import mfma_conv as MC

    prev = MC.fp32_mode()
    try:
        tr32, pcls, targets = _setup(512, 100.0, 1, 120000, torch.float32, seed=11)
        MC.set_fp32_mode("exact")
        tr16, _, _ = _setup(512, 100.0, 1, 120000, torch.bfloat16, seed=11)
        tr16.net.load_state_dict(tr32.net.state_dict())
        tr32.model.train(), tr16.model.train()
        with torch.no_grad():
            _, _, raw32, _ = tr32.net(None, pcls, None, decode=False)
            _, _, raw16, _ = tr16.net(None, pcls, None, decode=False)
        l32, _, _ = tr32.loss(pcls, targets)
        l16, _, _ = tr16.loss(pcls, targets)
    finally:
        MC.set_fp32_mode(prev)
    report = {}
    for h in HEADS:
        a, b = raw16[h].detach().double(), raw32[h].detach().double()
        report[h] = (float((a - b).abs().max() / b.abs().max()), float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()))
    print("bf16 vs exact fp32, per head (max / range, rms / rms):", {k: (f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in report.items()},
          f"loss {abs(float(l16) - float(l32)) / abs(float(l32)):.2e}")
    for h, (mx, rms) in report.items():
        assert mx <= 1e-1 and rms <= 8e-2, (h, mx, rms)
    assert abs(float(l16) - float(l32)) <= 5e-2 * abs(float(l32))


def test_full_size_step_properties():
    """BASELINE config 3 shape: 120k points, 512^2, B=2 -> finite loss, every parameter gets a finite gradient,
    two identical steps from the same state are bitwise identical (no float atomics in our kernels; MIOpen permitting
    we only require closeness there)."""
    tr, pcls, targets = _setup(512, 100.0, 2, 120000, torch.bfloat16, seed=7)
    tr.model.train()
    total, losses, _ = tr.loss(pcls, targets)
    total.backward()
    assert torch.isfinite(total)
    with torch.no_grad():  # the decoded dense predictions (inference path of BoxLearner.forward, simple_net.py:70-109)
        boxes, _, _, _ = tr.net(None, pcls, None)
    assert boxes.pos.shape == (2, 128 * 128, 3)
    for n, p in tr.net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_decode_loss_equals_torch_op_path(dtype):
    """include/liso_detector.h: activations + decode + CenterPoint loss + rotation regulariser in one pass must equal
    the torch-op mirrors of simple_net.py:111-151 / centerpoint_loss.py:13-136 / main_utils.py:51-58 -- every loss term,
    the total, and the gradients of all four raw network maps (NCHW fp32 and channels-last bf16-network layouts),
    with an ignore region and non-trivial rotation weights."""
    from liso_amd.losses.fused_centerpoint import fused_centerpoint_loss, supports

    tr, pcls, targets = _setup(128, 100.0, 2, 20000, dtype, seed=11)
    assert supports(tr.cfg) and tr.fused_loss
    tr.model.train()
    g = torch.Generator().manual_seed(3)
    ignore = (torch.rand(targets["center_bool_mask"].shape, generator=g) < 0.1).cuda()
    rot_w = (torch.rand(targets["probs"].shape, generator=g) * 2).cuda()
    _, _, raw, _ = tr.net(None, pcls, None, decode=False)
    raw = {k: (v.detach() * 3.0).requires_grad_(True) for k, v in raw.items()}  # spread the activations out
    gt_maps = {a: targets[a] for a in ("pos", "dims", "rot", "probs")}
    total, losses = fused_centerpoint_loss(cfg=tr.cfg, raw_box_maps=raw, gt_maps=gt_maps, gt_center_mask=targets["center_bool_mask"],
                                           ignore_region_is_true_mask=ignore, rotation_loss_weights_map=rot_w,
                                           pillar_center_coors_m=tr.net.pillar_center_coors_m)
    grads = torch.autograd.grad(total * 1.7, [raw[k] for k in ("pos", "dims", "rot", "probs")])
    # torch-op path on the same raw maps (fp64 for a clean comparison)
    from liso_amd.losses.centerpoint_loss import centerpoint_loss, rotation_vec_on_unit_circle
    raw64 = {k: v.detach().double().requires_grad_(True) for k, v in raw.items()}
    act = {k: tr.net.activations[k](v) for k, v in raw64.items()}
    from liso_amd.kabsch.output_modification import output_modification
    dec = output_modification({k: v.clone() for k, v in act.items()}, tr.cfg.box_prediction, tr.cfg.data, "boxes",
                              tr.net.pillar_center_coors_m.double())
    ref_losses = centerpoint_loss(loss_cfg=tr.cfg.loss, raw_activated_pred_box_maps=act, decoded_pred_box_maps=dec,
                                  gt_maps={k: v.double() for k, v in gt_maps.items()}, gt_center_mask=targets["center_bool_mask"],
                                  rotation_loss_weights_map=rot_w.double(), box_prediction_cfg=tr.cfg.box_prediction,
                                  ignore_region_is_true_mask=ignore)
    ref_total = sum(ref_losses.values()) * tr.cfg.loss.supervised.supervised_on_clusters.weight \
        + rotation_vec_on_unit_circle(act) * tr.cfg.box_prediction.rotation_representation.regul_weight
    ref_grads = torch.autograd.grad(ref_total * 1.7, [raw64[k] for k in ("pos", "dims", "rot", "probs")])
    for k, v in ref_losses.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-5 * max(abs(float(v)), 1e-3), (k, float(losses[k]), float(v))
    assert abs(float(total) - float(ref_total)) <= 1e-5 * abs(float(ref_total))
    for name, a, b in zip(("pos", "dims", "rot", "probs"), grads, ref_grads):
        assert a.stride() == raw[name].stride()
        assert _rel(a, b) < 1e-5, (name, _rel(a, b))


HEADS = ("pos", "dims", "rot", "probs")


def test_logit_maps_and_rpn_features_match_oracle_fp32():
    """north_star's tolerance is on the LOGITS: the raw per-head maps [B,128/4..,C] of the product (HIP pillar path, own
    convolution + BatchNorm kernels) within 1e-3 rel of the fp64 oracle, in train and in eval mode, and the BEV canvas that
    feeds the backbone (center_head.py:109-117, rpn.py:137-146)."""
    from oracle.train_step import detector_forward_loss, prepare_state

    for training in (True, False):
        tr, pcls, targets = _setup(128, 100.0, 2, 20000, seed=21)
        if not training:
            with torch.no_grad():
                for m in tr.net.modules():
                    if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                        m.running_mean.uniform_(-0.2, 0.2)
                        m.running_var.uniform_(0.6, 1.4)
        sd64 = prepare_state(tr.net.state_dict(), torch.float64)
        tr.net.train(training)
        with torch.no_grad():
            _, _, raw, aux = tr.net(None, pcls, None, decode=False)
            _, raw64, bev64 = detector_forward_loss(sd64, [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}, 128, 100.0,
                                                    training=training, dtype=torch.float64)
        for h in HEADS:
            assert raw[h].shape == raw64[h].shape, (h, raw[h].shape, raw64[h].shape)
            assert _rel(raw[h], raw64[h]) <= 1e-3, (training, h, _rel(raw[h], raw64[h]))


@pytest.mark.parametrize("fmode", ["exact", "x3"])
def test_rpn_head_on_gpu_match_reference_fixture(golden_dir, fmode):
    """`fmode` = arithmetic of the fp32 convolutions: "exact" (native fp32 MFMA, the reference's fp32 semantics: gradients at the
    limits the true-fp32 library path met before the F32X3 kernels existed) and "x3" (bf16 hi/lo pairs, the production default of
    the SLIM networks: bulk / worst-element limits).
    The reference's own RPN / CenterHead outputs (tests/golden/detector_rpn_head.npz, generated by importing
    liso/networks/centerpoint/{rpn,center_head}.py) reproduced ON THE GPU through the own convolution / BatchNorm kernels:
    feature map and every head's logits <= 1e-3, train and eval mode; gradients as in the CPU test of the same fixture."""
    import os

    from liso_amd.networks.centerpoint.center_head import CenterHead
    from liso_amd.networks.centerpoint.rpn import RPN

    from liso_amd.utils import mfma_conv as MC

    g = np.load(os.path.join(golden_dir, "detector_rpn_head.npz"))
    prev_mode = MC.set_fp32_mode(fmode)

    def sd(prefix):
        tag = "sd_" + prefix + "__"
        return {k[len(tag):].replace("__", "."): torch.from_numpy(g[k]).clone() for k in g.files if k.startswith(tag)}

    norm = {"affine": True, "track_running_stats": True}
    for tag, training in (("train", True), ("eval", False)):
        rpn = RPN(layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[16, 32, 64], us_layer_strides=[0.5, 1, 2],
                  us_num_filters=[32, 32, 32], num_input_features=16, norm_cfg=norm)
        head = CenterHead(common_heads={"pos": (3, 2), "dims": (3, 2), "rot": (2, 2), "probs": (1, 2)}, norm_cfg=norm,
                          in_channels=96, stride=1, share_conv_channel=16)
        rpn.load_state_dict(sd("rpn"), strict=True), head.load_state_dict(sd("head"), strict=True)
        rpn.cuda().train(training), head.cuda().train(training)
        x = torch.from_numpy(g["x"]).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        feat = rpn(x)
        pred = head(feat)
        assert _rel(feat, torch.from_numpy(g[f"{tag}_feat"])) <= 1e-3
        for h in HEADS:
            assert _rel(pred[h], torch.from_numpy(g[f"{tag}_{h}"])) <= 1e-3, (tag, h)
        if training:
            sum((v * torch.linspace(-1, 1, v.numel(), device="cuda").view_as(v)).sum() for v in pred.values()).backward()
            # fp32 tensors run as bf16 hi/lo pairs (forward error 1e-5 instead of 1e-7): a ReLU input within that distance of
            # zero may open on one side only, and through the train-mode BatchNorm statistics that one pixel moves every
            # gradient upstream a little (tests/test_gpu_conv.py checks the smooth case at 1e-3) -> the bulk
            # of each gradient tensor is held tightly, its worst element loosely
            def bulk(a, b):
                a, b = a.detach().cpu().double().numpy(), np.asarray(b, np.float64)
                return np.median(np.abs(a - b)) / np.median(np.abs(b))

            for got, key in ((x.grad, "train_grad_x"), (rpn.blocks[0][1].weight.grad, "train_grad_rpn_blocks_0_1_weight"),
                             (rpn.deblocks[2][0].weight.grad, "train_grad_rpn_deblocks_2_0_weight")):
                lim_bulk, lim_worst = (1e-3, 4e-3) if fmode == "exact" else (5e-2, 0.3)  # (8x8 maps: 128 values per BN statistic)
                assert bulk(got, g[key]) <= lim_bulk and _rel(got, torch.from_numpy(g[key])) <= lim_worst, \
                    (fmode, key, bulk(got, g[key]), _rel(got, torch.from_numpy(g[key])))
            # (the fixture's upstream gradient sums to ~0 over the map: this bias gradient is pure rounding, |g| = 3e-4)
            assert abs(head.tasks[0].probs[3].bias.grad.reshape(-1)[0].item() - float(np.asarray(g["train_grad_head_probs_3_bias"]).reshape(-1)[0])) <= 2e-3
            assert _rel(rpn.blocks[0][2].running_mean, torch.from_numpy(g["train_rm_after_rpn_blocks_0_2"])) <= 1e-3
    MC.set_fp32_mode(prev_mode)


def test_config5_train_step_300k_points_1024_grid_bf16():
    """BASELINE configs[4]: nuScenes-shaped 300k-point clouds (5 channels: x, y, z, intensity, time), 1024 x 1024 BEV,
    reduced precision.  The reference has no AMP at all; the build's reduced-precision type is bf16 (same MFMA
    rate as fp16 on gfx950, fp32 range: no loss scaling), stated in DESIGN.md.  One full train step: finite loss, finite
    gradient for every parameter, and the bf16 loss tracks the fp32 loss of the same step."""
    from liso_amd.datasets.synthetic import detector_batch
    from liso_amd.trainer import DetectorTrainer
    from liso_amd.utils.config import default_cfg

    dev = torch.device("cuda:0")
    losses = {}
    for dtype in (torch.bfloat16, torch.float32):
        torch.manual_seed(5)
        cfg = default_cfg(grid=1024, bev_range_m=100.0)
        cfg.data.num_point_channels = 5
        tr = DetectorTrainer(cfg, dev, compute_dtype=dtype, total_steps=8)
        pcls, targets = detector_batch(9, 1, dev, n_points=300000, grid=1024, bev_range_m=100.0)
        gen = torch.Generator().manual_seed(3)  # 5th channel: sweep time offset in [0, 0.5) s
        pcls = [torch.cat([p, (torch.randint(0, 10, (p.shape[0], 1), generator=gen).float() * 0.05).to(dev)], dim=1) for p in pcls]
        assert pcls[0].shape == (300000, 5)
        tr.model.train()
        total, _, _ = tr.loss(pcls, targets)
        total.backward()
        assert torch.isfinite(total)
        for n, p in tr.net.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
        losses[dtype] = float(total)
        if dtype == torch.bfloat16:
            l0, l1 = float(tr.step(pcls, targets)), float(tr.step(pcls, targets))  # two full optimizer steps
            assert np.isfinite(l0) and np.isfinite(l1)
        del tr
        torch.cuda.empty_cache()
    assert abs(losses[torch.bfloat16] - losses[torch.float32]) <= 5e-2 * abs(losses[torch.float32]), losses


def test_two_graph_step_with_the_backward_pass_cut_behind_block0_equals_the_one_graph_step():
    """DetectorTrainer(use_graph=True, grad_buckets=2): the step is captured as two hipGraphs cut behind the backbone's first block
    (mfma_conv.GradCut; several ranks all-reduce the first graph's gradients while the second one runs).  Same arithmetic as the
    one-graph step: losses and parameters after 4 steps bit for bit."""
    from liso_amd.datasets.synthetic import detector_batch
    from liso_amd.trainer import DetectorTrainer
    from liso_amd.utils.config import default_cfg

    dev = torch.device("cuda:0")
    pcls, targets = detector_batch(44, 2, dev, n_points=30000, grid=256, bev_range_m=50.0)
    out = []
    for buckets in (1, 2):
        torch.manual_seed(3)
        tr = DetectorTrainer(default_cfg(grid=256, bev_range_m=50.0), dev, compute_dtype=torch.bfloat16, total_steps=12, use_graph=True,
                             grad_buckets=buckets)
        losses = [float(tr.step(pcls, targets)) for _ in range(4)]
        assert tr.n_grad_buckets == buckets and (tr._graph2 is not None) == (buckets == 2)
        out.append((losses, torch.cat([p.detach().float().flatten() for p in tr.net.parameters()]).cpu()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_merged_output_convolutions_of_the_heads_equal_the_four_separate_ones(dtype):
    """SepHead.forward_fused runs the four output convolutions (64 -> 3 / 3 / 2 / 1 on their slices of the merged hidden map) as ONE
    convolution with block-diagonal filters (center_head.py:_BlockDiagonalFilters).  Zeros off the diagonal add exact zeros: logits
    and every gradient equal the four-launch path to fp32 summation order."""
    tr, pcls, targets = _setup(128, 100.0, 2, 20000, dtype, seed=13)
    head = tr.net.model.center_head.tasks[0]
    res = []
    for merged in (False, True):
        head.merge_output_convs = merged
        tr.net.zero_grad(set_to_none=True)
        tr.optimizer.zero_grad()
        tr.model.train()
        with torch.no_grad():  # (same BatchNorm buffers for both passes)
            buf = {k: v.clone() for k, v in tr.net.state_dict().items()}
        _, _, raw, _ = tr.net(None, pcls, None, decode=False)
        loss = sum((v.float() * torch.linspace(-1, 1, v.numel(), device=v.device).view_as(v)).sum() for v in raw.values())
        loss.backward()
        res.append(({k: v.detach().float().clone() for k, v in raw.items()},
                    {n: p.grad.detach().float().clone() for n, p in tr.net.named_parameters() if p.grad is not None and "center_head" in n}))
        tr.net.load_state_dict(buf)
    head.merge_output_convs = True
    for h in HEADS:
        assert res[0][0][h].shape == res[1][0][h].shape and _rel(res[1][0][h], res[0][0][h]) <= 2e-6, (h, _rel(res[1][0][h], res[0][0][h]))
    assert set(res[0][1]) == set(res[1][1]) and len(res[0][1]) >= 20
    for n in res[0][1]:
        if n.endswith(".0.bias"):  # a convolution bias in front of a BatchNorm: its true gradient is 0, both values are rounding noise
            continue
        lim = 2e-2 if dtype == torch.bfloat16 else 1e-4  # (bf16: the data gradient of the hidden map is rounded once more or less)
        assert _rel(res[1][1][n], res[0][1][n]) <= lim, (n, _rel(res[1][1][n], res[0][1][n]))
