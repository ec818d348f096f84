"""Fused CenterPoint decode + loss (include/liso_detector.h) behind the reference's loss vocabulary.

`fused_centerpoint_loss(...)` returns the same dictionary keys as `centerpoint_loss` (liso/losses/centerpoint_loss.py:13-136)
plus the weighted total including the rotation regulariser (liso/kabsch/main_utils.py:119-134), computed from the RAW
network maps: activations (simple_net_utils.py:8-14) and decode (output_modification.py:4-45) happen inside the kernel.
`supports(cfg)` tells whether a configuration is the CenterPoint-pillar overlay the kernel implements; other
configurations keep the torch-op path of liso_amd.losses.centerpoint_loss."""
import ctypes

import torch

from liso_amd import _lib as L

KEYS = ("loss/supervised/centermaps/probs", "loss/supervised/centermaps/rot", "loss/supervised/centermaps/dims",
        "loss/supervised/centermaps/pos")


def supports(cfg) -> bool:
    bp = cfg.box_prediction
    a, om = bp.activations, bp.output_modification
    sup = cfg.loss.supervised
    return (a.pos == "tanh" and a.dims == "softplus" and a.rot == "none" and a.probs == "none"
            and bp.position_representation.method == "local_relative_offset" and bp.position_representation.num_box_pos_dims == 3
            and bp.rotation_representation.method == "vector" and not bp.rotation_representation.norm_vector_len
            and bp.dimensions_representation.method == "predict_abs_size"
            and all(om[k] == "none" for k in ("pos", "dims", "rot", "probs"))
            and sup.centermaps.confidence_target == "gaussian"
            and tuple(sup.supervised_on_clusters.attrs) == ("pos", "dims", "rot", "probs"))


def _strides(*maps):
    out = []
    for m in maps:  # logical [B,H,W,C] views of any memory layout -> (batch, channel, row, column) strides
        sb, sh, sw, sc = m.stride()
        out += [sb, sc, sh, sw]
    return (ctypes.c_long * 16)(*out)


class _CenterLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, dims, rot, probs, ccfg, gt, center_mask, ignore_mask, rot_weights, centers):
        L.require_cuda(pos, dims, rot, probs)
        assert pos.dtype == dims.dtype == rot.dtype == probs.dtype == torch.float32
        dev = pos.device
        lib = L.lib()
        sums = torch.empty(12, dtype=torch.float64, device=dev)
        losses = torch.empty(6, dtype=torch.float32, device=dev)
        nbytes = lib.liso_centerloss_workspace_bytes(ctypes.byref(ccfg))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        tens = (gt["probs"].float().contiguous(), gt["dims"].float().contiguous(), gt["pos"].float().contiguous(),
                gt["rot"].float().contiguous(), center_mask.to(torch.uint8).contiguous(),
                None if ignore_mask is None else ignore_mask.to(torch.uint8).contiguous(),
                None if rot_weights is None else rot_weights.float().contiguous(), centers.float().contiguous())
        st = _strides(pos, dims, rot, probs)
        opt = lambda t: L.ptr(t) if t is not None else None  # noqa: E731
        with torch.cuda.device(dev):
            L.check(L.TIMER.launch("centerloss_fwd", lambda: lib.liso_centerloss_fwd_f32(
                ctypes.byref(ccfg), L.ptr(pos), L.ptr(dims), L.ptr(rot), L.ptr(probs), st, L.ptr(tens[0]), L.ptr(tens[1]),
                L.ptr(tens[2]), L.ptr(tens[3]), L.ptr(tens[4]), opt(tens[5]), opt(tens[6]), L.ptr(tens[7]), L.ptr(sums),
                L.ptr(losses), L.ptr(ws), nbytes, L.stream_ptr())), "centerloss_fwd")
        ctx.save_for_backward(pos, dims, rot, probs, sums, *[t for t in tens if t is not None])
        ctx.has = (tens[5] is not None, tens[6] is not None)
        ctx.ccfg = ccfg
        ctx.mark_non_differentiable(losses)
        return losses[5], losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        pos, dims, rot, probs, sums, *rest = ctx.saved_tensors
        gtp, gtd, gtpos, gtr, cm = rest[:5]
        rest = list(rest[5:])
        ign = rest.pop(0) if ctx.has[0] else None
        rw = rest.pop(0) if ctx.has[1] else None
        centers = rest.pop(0)
        grads = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device) for t in (pos, dims, rot, probs)]
        g = g_total.float().reshape(1).contiguous()
        opt = lambda t: L.ptr(t) if t is not None else None  # noqa: E731
        with torch.cuda.device(pos.device):
            L.check(L.TIMER.launch("centerloss_bwd", lambda: L.lib().liso_centerloss_bwd_f32(
                ctypes.byref(ctx.ccfg), L.ptr(pos), L.ptr(dims), L.ptr(rot), L.ptr(probs), _strides(pos, dims, rot, probs),
                L.ptr(gtp), L.ptr(gtd), L.ptr(gtpos), L.ptr(gtr), L.ptr(cm), opt(ign), opt(rw), L.ptr(centers), L.ptr(sums),
                L.ptr(g), L.ptr(grads[0]), L.ptr(grads[1]), L.ptr(grads[2]), L.ptr(grads[3]), L.stream_ptr())), "centerloss_bwd")
        return (*grads, None, None, None, None, None, None)


def fused_centerpoint_loss(*, cfg, raw_box_maps, gt_maps, gt_center_mask, ignore_region_is_true_mask=None,
                           rotation_loss_weights_map=None, pillar_center_coors_m):
    """raw_box_maps: dict pos/dims/rot/probs of logical [B,H,W,C] fp32 network outputs (any strides).
    -> (total, losses): total = weight * sum(losses) + regul_weight * rotation_vec_on_unit_circle, differentiable;
    losses = {reference keys: detached scalars}"""
    B, H, W, _ = raw_box_maps["pos"].shape
    pr = cfg.box_prediction.position_representation
    rr = cfg.box_prediction.rotation_representation
    reg_w = rr.regul_weight if rr.get("regularization", None) == "rot_vec_on_unit_circle" or "regul_weight" in rr else 0.0
    ccfg = L.CenterLossCfg(B, H, W, float(cfg.data.bev_range_m[0]) / H, float(cfg.data.bev_range_m[1]) / W,
                           float(pr.box_z_pos_prior_min), float(pr.box_z_pos_prior_max),
                           float(cfg.loss.supervised.supervised_on_clusters.weight), float(reg_w))
    rw = None if rotation_loss_weights_map is None else rotation_loss_weights_map.reshape(B, H, W)
    total, losses = _CenterLoss.apply(raw_box_maps["pos"], raw_box_maps["dims"], raw_box_maps["rot"], raw_box_maps["probs"], ccfg,
                                      gt_maps, gt_center_mask, ignore_region_is_true_mask, rw, pillar_center_coors_m)
    out = {k: losses[i] for i, k in enumerate(KEYS)}
    out["loss/regularization/rot_vec_on_unit_circle"] = losses[4]
    return total, out
