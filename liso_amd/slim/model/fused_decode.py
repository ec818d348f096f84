"""Host side of include/liso_slim_decode.h: the per-point training decoder of SLIM and its point-wise loss terms as a handful of
launches with hand-written backward passes.

What they replace (reference): HeadDecoder.apply_output_modification / apply_flow_to_points (liso/slim/model/head_decoder.py:67-408,
517-955) on the rows of every point, the weights / warped cloud of the static aggregation (slim_loss/static_aggregation.py:34-68),
static_points_loss (slim_loss_adaptor.py:55-91) and the nearest-point loss bookkeeping (knn_loss.py:9-82, knn_wrapper.py:155-217).
`HeadDecoder._forward_pointwise` and `selfsupervisedSlimSingleScaleLoss` call in here when the configuration is covered
(`decode_supported`); everything else keeps the torch formulation of the same arithmetic.
"""
import ctypes

import torch

from liso_amd import _lib as L

_MODES = {"net": 0, True: 1, False: 2}


def decode_supported(cfg, network_output):
    """network channels or constants only: the `gt_*` output modes read dataset labels and stay on the torch formulation"""
    om = cfg.model.output_modification
    ok_logits = all((v is True or v is False or v == "net") for v in (om.disappearing_logit, om.static_logit, om.dynamic_logit, om.ground_logit))
    ok_flows = om.static_flow in ("net", "zero") and om.dynamic_flow in ("net", "zero")
    return (ok_logits and ok_flows and network_output.is_cuda and network_output.dtype == torch.float32 and network_output.shape[-1] == 8
            and cfg.model.predict_weight_for_static_aggregation is False and float(om.dynamic_flow_grad_scale) >= 0.0)


def _mode(v):
    return 0 if v == "net" else (1 if v is True else 2)


class DecodeMeta:
    """everything of one decode call that is not differentiable: configuration, cells of the points, filled-pillar mask, BEV-wide
    logit extrema, threshold, cloud"""

    def __init__(self, cfg, bev_extent, plan, filled_pillar_mask, extrema, threshold, pc, *, overwrite_flow=True, overwrite_logits=True,
                 non_rigid=False):
        S, N, H, W = plan.shape
        om = cfg.model.output_modification
        # the reference's consistency checks (head_decoder.py:812-817, :827-829, :905-907)
        if om.static_logit is True:
            assert om.dynamic_logit is False and om.ground_logit is False
        if om.static_logit is False:
            assert om.dynamic_logit is not False or om.ground_logit is not False
        if om.ground_logit is True:
            assert om.static_logit is False and om.dynamic_logit is False
        c = L.SlimDecodeCfg()
        c.samples, c.n, c.h, c.w = S, N, H, W
        for i, v in enumerate((om.disappearing_logit, om.static_logit, om.dynamic_logit, om.ground_logit)):
            c.logit_mode[i] = _mode(v)
        c.static_flow_zero, c.dynamic_flow_zero = int(om.static_flow == "zero"), int(om.dynamic_flow == "zero")
        c.overwrite_flow, c.overwrite_logits = int(overwrite_flow), int(overwrite_logits)
        c.non_rigid, c.use_static_aggr = int(bool(non_rigid)), int(bool(cfg.model.use_static_aggr_flow_for_aggr_flow))
        c.dyn_grad_scale = float(om.dynamic_flow_grad_scale)
        e = [float(v) for v in bev_extent]
        c.ext_lo[0], c.ext_lo[1], c.ext_span[0], c.ext_span[1] = e[0], e[1], e[2] - e[0], e[3] - e[1]
        self.cfg, self.shape = c, (S, N, H, W)
        dev = pc.device
        self.lin = plan.lin
        self.filled = filled_pillar_mask.contiguous()
        assert self.filled.dtype == torch.bool and self.filled.numel() == S * H * W
        self.extrema = None if extrema is None else torch.cat([extrema[0].reshape(4), extrema[1].reshape(4)]).float().contiguous()
        thr = threshold if torch.is_tensor(threshold) else torch.tensor(float(threshold), device=dev)
        self.threshold = thr.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        self.pc = pc.detach()
        assert self.pc.dtype == torch.float32 and self.pc.is_contiguous() and self.pc.shape[:2] == (S, N)


class _DecodeWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, meta):
        S, N, _, _ = meta.shape
        raw = raw.contiguous()
        dev = raw.device
        x = torch.empty((S, N, 3), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        w = torch.empty((S, N), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().liso_slim_decode_weights_fwd(
                ctypes.byref(meta.cfg), L.ptr(raw), L.ptr(meta.lin), L.ptr(meta.filled),
                L.ptr(meta.extrema) if meta.extrema is not None else None, L.ptr(meta.pc), meta.pc.shape[-1], L.ptr(x), L.ptr(y), L.ptr(w),
                L.stream_ptr()), "slim_decode_weights_fwd")
        ctx.save_for_backward(raw)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(x)
        return x, y, w

    @staticmethod
    def backward(ctx, gx, gy, gw):
        (raw,) = ctx.saved_tensors
        meta = ctx.meta
        gy = None if gy is None else gy.float().contiguous()
        gw = None if gw is None else gw.float().contiguous()
        graw = torch.empty_like(raw)
        with torch.cuda.device(raw.device):
            L.check(L.lib().liso_slim_decode_weights_bwd(
                ctypes.byref(meta.cfg), L.ptr(raw), L.ptr(meta.lin), L.ptr(meta.filled),
                L.ptr(meta.extrema) if meta.extrema is not None else None, L.ptr(gy) if gy is not None else None,
                L.ptr(gw) if gw is not None else None, L.ptr(graw), L.stream_ptr()), "slim_decode_weights_bwd")
        return graw, None


_OUT = L.SlimDecodeOut.FIELDS
_OUT_SHAPES = {"dis_logit": (), "dis": (), "logits": (3,), "probs": (3,), "staticness": (), "dynamicness": (), "groundness": (),
               "dyn_flow": (3,), "stat_flow": (3,), "agg_flow": (3,), "saf_flow": (3,)}


class _DecodePoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, trafo, meta, want=None):
        S, N, _, _ = meta.shape
        raw = raw.contiguous()
        T = trafo.detach().double().contiguous()
        dev = raw.device
        outs = {k: (torch.empty((S, N) + _OUT_SHAPES[k], dtype=torch.float32, device=dev) if want is None or k in want else None) for k in _OUT}
        flags = torch.empty((S, N, 3), dtype=torch.bool, device=dev) if want is None else None
        o = L.SlimDecodeOut(*[L.ptr(outs[k]) if outs[k] is not None else None for k in _OUT], L.ptr(flags) if flags is not None else None)
        with torch.cuda.device(dev):
            L.check(L.lib().liso_slim_decode_points_fwd(
                ctypes.byref(meta.cfg), L.ptr(raw), L.ptr(meta.lin), L.ptr(meta.filled),
                L.ptr(meta.extrema) if meta.extrema is not None else None, L.ptr(meta.threshold), L.ptr(T), ctypes.byref(o),
                L.stream_ptr()), "slim_decode_points_fwd")
        ctx.save_for_backward(raw, T)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        if flags is not None:
            ctx.mark_non_differentiable(flags)
        return tuple(outs[k] for k in _OUT) + (flags,)

    @staticmethod
    def backward(ctx, *grads):
        raw, T = ctx.saved_tensors
        meta = ctx.meta
        S, N, H, W = meta.shape
        g = [None if t is None else t.float().contiguous() for t in grads[:len(_OUT)]]
        gd = dict(zip(_OUT, g))
        want_T = ctx.needs_input_grad[1] and (gd["saf_flow"] is not None or (meta.cfg.use_static_aggr and gd["agg_flow"] is not None))
        graw = torch.empty_like(raw)
        gsaf = torch.empty((S, N, 2), dtype=torch.float32, device=raw.device) if want_T else None
        go = L.SlimDecodeOut(*[L.ptr(t) if t is not None else None for t in g], None)
        with torch.cuda.device(raw.device):
            L.check(L.lib().liso_slim_decode_points_bwd(
                ctypes.byref(meta.cfg), L.ptr(raw), L.ptr(meta.lin), L.ptr(meta.filled),
                L.ptr(meta.extrema) if meta.extrema is not None else None, L.ptr(meta.threshold), L.ptr(T), ctypes.byref(go), L.ptr(graw),
                L.ptr(gsaf) if gsaf is not None else None, L.stream_ptr()), "slim_decode_points_bwd")
        gT = None
        if want_T:  # saf = (T - I)[:2] (cx, cy, 0, 1): d/dT[i, 0] = sum g_i cx, d/dT[i, 1] = sum g_i cy, d/dT[i, 3] = sum g_i  (rare configurations)
            cell = meta.lin.view(S, N).clamp(min=0).long() % (H * W)
            c = meta.cfg
            cx = ((torch.div(cell, W, rounding_mode="floor").double() + 0.5) / H) * c.ext_span[0] + c.ext_lo[0]
            cy = (((cell % W).double() + 0.5) / W) * c.ext_span[1] + c.ext_lo[1]
            g64 = gsaf.double()
            gT = torch.zeros((S, 4, 4), dtype=torch.float64, device=raw.device)
            gT[:, :2, 0] = (g64 * cx[..., None]).sum(dim=1)
            gT[:, :2, 1] = (g64 * cy[..., None]).sum(dim=1)
            gT[:, :2, 3] = g64.sum(dim=1)
        return graw, gT, None, None


def decode_points(raw, trafo, meta, want=None):
    """-> dict of the per-point predictions (float tensors + `flags` bool [S,N,3]); `want`: the subset of outputs to write (inference)"""
    res = _DecodePoints.apply(raw, trafo, meta, want)
    out = dict(zip(_OUT, res[:-1]))
    out["flags"] = res[-1]
    return out


decode_weights = _DecodeWeights.apply


# ---- losses ----------------------------------------------------------------------------------------------------------------------
def _u8(mask):
    return mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()


class _StaticPointsLossMean(torch.autograd.Function):
    """masked mean over the valid rows of static_points_loss (slim_loss_adaptor.py:55-91): one reduction launch forward, one
    elementwise launch backward"""

    @staticmethod
    def forward(ctx, pc, valid, flow, weight, trafo):
        S, N = valid.shape
        pc, flow, weight = pc.detach().float().contiguous(), flow.float().contiguous(), weight.float().contiguous()
        v8, T = _u8(valid), trafo.detach().double().contiguous()
        lib = L.lib()
        nbytes = lib.liso_slim_loss_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pc.device)
        out = torch.empty(1, dtype=torch.float32, device=pc.device)
        with torch.cuda.device(pc.device):
            L.check(lib.liso_slim_static_points_loss_fwd(S, N, L.ptr(pc), pc.shape[-1], L.ptr(v8), L.ptr(flow), L.ptr(weight), L.ptr(T),
                                                         L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr()), "slim_static_points_loss_fwd")
        ctx.save_for_backward(pc, v8, flow, weight, T, ws)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        pc, v8, flow, weight, T, ws = ctx.saved_tensors
        S, N = v8.shape
        g = g.float().reshape(1).contiguous()
        gf = torch.empty_like(flow) if ctx.needs_input_grad[2] else None
        gw = torch.empty_like(weight) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(pc.device):
            L.check(L.lib().liso_slim_static_points_loss_bwd(
                S, N, L.ptr(pc), pc.shape[-1], L.ptr(v8), L.ptr(flow), L.ptr(weight), L.ptr(T), L.ptr(g), L.ptr(ws),
                L.ptr(gf) if gf is not None else None, L.ptr(gw) if gw is not None else None, L.stream_ptr()),
                "slim_static_points_loss_bwd")
        return None, None, gf, gw, None


static_points_loss_mean = _StaticPointsLossMean.apply


class _NearestPointLossMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pc, valid, flow, cloud_b, index, order, cfg):
        S, N = valid.shape
        pc, flow, cloud_b = pc.detach().float().contiguous(), flow.float().contiguous(), cloud_b.detach().float().contiguous()
        v8 = _u8(valid)
        lib = L.lib()
        nbytes = lib.liso_slim_loss_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pc.device)
        out = torch.empty(1, dtype=torch.float32, device=pc.device)
        d2 = torch.empty((S, N), dtype=torch.float32, device=pc.device)
        with torch.cuda.device(pc.device):
            L.check(lib.liso_slim_nearest_point_loss_fwd(
                ctypes.byref(cfg), L.ptr(pc), pc.shape[-1], L.ptr(v8), L.ptr(flow), L.ptr(cloud_b), cloud_b.shape[-1], L.ptr(index),
                L.ptr(order) if order is not None else None, L.ptr(d2), L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr()),
                "slim_nearest_point_loss_fwd")
        ctx.save_for_backward(pc, v8, flow, cloud_b, index, ws)
        ctx.order, ctx.cfg = order, cfg
        ctx.set_materialize_grads(False)
        return out.reshape(()), d2

    @staticmethod
    def backward(ctx, g, gd2):
        pc, v8, flow, cloud_b, index, ws = ctx.saved_tensors
        g = (torch.zeros(1, dtype=torch.float32, device=pc.device) if g is None else g.float().reshape(1)).contiguous()
        gd2 = None if gd2 is None else gd2.float().contiguous()
        gf = torch.empty_like(flow)
        with torch.cuda.device(pc.device):
            L.check(L.lib().liso_slim_nearest_point_loss_bwd(
                ctypes.byref(ctx.cfg), L.ptr(pc), pc.shape[-1], L.ptr(v8), L.ptr(flow), L.ptr(cloud_b), cloud_b.shape[-1], L.ptr(index),
                L.ptr(ctx.order) if ctx.order is not None else None, L.ptr(g), L.ptr(gd2) if gd2 is not None else None, L.ptr(ws),
                L.ptr(gf), L.stream_ptr()), "slim_nearest_point_loss_bwd")
        return None, None, gf, None, None, None, None


def knn_losses(pc_a, valid_a, cloud_b, flows, knn_indices, query_order_indices, *, bev_extent, knn_loss_cfg):
    """The nearest-point loss of several flow types of one direction (knn_loss.py:9-82 + knn_wrapper.py:155-217 + the masked mean
    of slim_loss_adaptor.py:239-251) -> {type: {"mean": 0-d loss, "knn": {nearest_dist_sqr [S,N] (0 at padding rows)}}}, or None
    when the call shape is not covered (the caller then uses the torch formulation).
    pc_a [S,N,>=3], valid_a [S,N], cloud_b [S,n_b,>=3]; flows {type: [S,N,3]}; knn_indices: S device indices stacked
    [iteration][cloud] (the same `clouds` objects repeating); query_order_indices: likewise for the rows of pc_a, or None."""
    from liso_amd.slim.slim_loss.knn_graph import KnnIndex, knn_graph
    from liso_amd.slim.slim_loss.knn_wrapper import NearestPointLoss

    S, N = valid_a.shape
    if (knn_indices is None or not pc_a.is_cuda or knn_loss_cfg.drop_outliers__perc != 0.0
            or knn_loss_cfg.fov_mode not in NearestPointLoss.FOV_MODES or len(flows) > 8):
        return None
    ids = [id(k) for k in knn_indices]
    clouds = len(set(ids))
    if S % clouds or any(ids[s] != ids[s % clouds] for s in range(S)) or not all(isinstance(k, KnnIndex) for k in knn_indices[:clouds]):
        return None
    order = None
    if query_order_indices is not None and all(id(query_order_indices[s]) == id(query_order_indices[s % clouds]) for s in range(S)):
        per = [getattr(query_order_indices[b], "sorted_ids", lambda: None)() for b in range(clouds)]
        if all(p is not None and p.shape[0] == N for p in per):
            order = per[0] if clouds == 1 else torch.stack(per, dim=0)
            order = order.contiguous()
    types = sorted(flows)
    pc = pc_a.detach().float().contiguous()
    v8 = _u8(valid_a)
    fl = [flows[t].detach().float().contiguous() for t in types]
    dev = pc.device
    query = torch.empty((len(types), S, N, 3), dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * len(types))(*[L.ptr(f) for f in fl])
    with torch.cuda.device(dev):
        L.check(L.lib().liso_slim_knn_queries(S, clouds, N, len(types), L.ptr(pc), pc.shape[-1], L.ptr(v8), ptrs,
                                              L.ptr(order) if order is not None else None, L.ptr(query), L.stream_ptr()),
                "slim_knn_queries")
    if clouds == 1:
        index = knn_graph(query.view(-1, 3), index=knn_indices[0], k=1, loop=True).view(len(types), S, N)
    else:
        index = torch.empty((len(types), S, N), dtype=torch.int64, device=dev)
        for b in range(clouds):
            q = query[:, b::clouds].reshape(-1, 3)
            index[:, b::clouds] = knn_graph(q, index=knn_indices[b], k=1, loop=True).view(len(types), S // clouds, N)
    e = [float(v) for v in bev_extent]
    cfg = L.SlimNpLossCfg(S, clouds, N, cloud_b.shape[1], (ctypes.c_float * 4)(*e[:4]), NearestPointLoss.FOV_MODES[knn_loss_cfg.fov_mode],
                          float(knn_loss_cfg.L1_delta))
    out = {}
    for i, t in enumerate(types):
        mean, d2 = _NearestPointLossMean.apply(pc, valid_a, flows[t], cloud_b, index[i], order, cfg)
        out[t] = {"mean": mean, "knn": {"nearest_dist_sqr": d2}}
    return out


def trafo_distance_from_moments(delta_trafos, points, mask):
    """trafo_distance (slim_loss_adaptor.py:9-33): mean over the masked points of |delta[:3, :] (p, 1)|^2, from the second moments
    of the cloud (one pass over the points, independent of the transforms) instead of transforming every point:
    sum_k |D p_k|^2 = tr(D M D^T), M = sum_k (p_k, 1)(p_k, 1)^T."""
    from liso_amd.slim.slim_loss.weighted_pc_alignment import _WeightedMoments

    p = torch.where(mask[..., None], points.detach()[..., :3].float(), 0.0)
    mom = _WeightedMoments.apply(p, p, mask.float())  # [S,16] fp64: sum w | sum w p | sum w p | sum w p p^T
    S = mom.shape[0]
    M = torch.empty((S, 4, 4), dtype=torch.float64, device=mom.device)
    M[:, :3, :3] = mom[:, 7:16].view(S, 3, 3)
    M[:, :3, 3] = mom[:, 1:4]
    M[:, 3, :3] = mom[:, 1:4]
    M[:, 3, 3] = mom[:, 0]
    D = delta_trafos[..., :3, :].double()
    val = torch.einsum("bij,bjk,bik->b", D, M, D)
    return (val / mom[:, 0]).float()
