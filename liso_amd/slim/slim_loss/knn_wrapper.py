"""Nearest-point flow loss.  Mirror of liso/slim/slim_loss/knn_wrapper.py:11-217 (same functions and semantics)."""
import ctypes
import functools as ft

import torch

from liso_amd import _lib as L

from liso_amd.slim.slim_loss.knn_graph import KnnIndex, knn_graph
from liso_amd.utils.config import AttrDict as Munch


def huber_delta(*, err=None, err_sqr=None, delta: float, mode: str = "large_grad_1"):
    """reference :11-49"""
    assert mode in {"large_grad_1", "small_err_sqr"}
    if delta == 0.0:
        assert mode == "large_grad_1" and err is None and err_sqr is not None
        nz = ~(err_sqr == 0.0)  # gradient-safe sqrt
        return torch.where(nz, err_sqr, torch.ones_like(err_sqr)).sqrt() * nz.to(torch.float)
    assert delta > 0.0
    if err is not None:
        assert err_sqr is None
        err_sqr = err.square()
    if mode == "large_grad_1":
        return torch.clamp(err_sqr, max=delta ** 2) / (2.0 * delta) + torch.clamp(err_sqr, min=delta ** 2).sqrt() - delta
    return torch.clamp(err_sqr, max=delta ** 2) + torch.clamp(err_sqr, min=delta ** 2).sqrt() * (2 * delta) - 2 * delta ** 2


def squared_sum(delta, dim: int = -1):
    return delta.square().sum(dim=dim)


class _NearestPointLossFused(torch.autograd.Function):
    """gather + squared distance + field-of-view weight + huber norm in one launch (include/liso_slim.h)"""

    @staticmethod
    def forward(ctx, cloud_a, flow, cloud_b, idx, cfg):
        a, f, b = cloud_a.float().contiguous(), flow.float().contiguous(), cloud_b.float().contiguous()
        i = idx.reshape(idx.shape[0], idx.shape[1]).contiguous()
        loss = torch.empty(a.shape[:2], dtype=torch.float32, device=a.device)
        d2 = torch.empty_like(loss)
        with torch.cuda.device(a.device):
            L.check(L.TIMER.launch("nearest_point_loss_fwd", lambda: L.lib().liso_nearest_point_loss_fwd_f32(
                ctypes.byref(cfg), L.ptr(a), L.ptr(f), L.ptr(b), L.ptr(i), L.ptr(loss), L.ptr(d2), L.stream_ptr())),
                "nearest_point_loss_fwd")
        ctx.save_for_backward(a, f, b, i)
        ctx.cfg = cfg
        return loss, d2

    @staticmethod
    def backward(ctx, g_loss, g_d2):
        a, f, b, i = ctx.saved_tensors
        gl = g_loss.float().contiguous()
        gd = None if g_d2 is None else g_d2.float().contiguous()
        gf = torch.empty_like(f)
        with torch.cuda.device(a.device):
            L.check(L.lib().liso_nearest_point_loss_bwd_f32(ctypes.byref(ctx.cfg), L.ptr(a), L.ptr(f), L.ptr(b), L.ptr(i), L.ptr(gl),
                                                            L.ptr(gd) if gd is not None else None, L.ptr(gf), L.stream_ptr()),
                    "nearest_point_loss_bwd")
        return (gf if ctx.needs_input_grad[0] else None), gf, None, None, None


class NearestPointLoss:
    """reference :58-135"""

    FOV_MODES = {"none": 0, "ignore_out_fov": 1, "mask_close_fov": 2}

    def __init__(self, *args, bev_extent, L1_delta: float, drop_outliers__perc: float, fov_mode: str = "ignore_out_fov", **kwargs):
        assert 0.0 <= drop_outliers__perc < 100.0
        assert fov_mode in {"none", "ignore_out_fov", "use_nearest", "mask_close_fov"}
        self.bev_extent = bev_extent
        self.L1_delta = L1_delta
        self.drop_outliers__perc = drop_outliers__perc
        self.huber_loss = ft.partial(huber_delta, delta=L1_delta, mode="large_grad_1")
        self.fov_mode = fov_mode

    def __call__(self, *, cloud_b__a, nearest_cloud_b__a, nearest_dist_sqr_b__a):
        e = self.bev_extent
        min_fov = torch.min(torch.stack([cloud_b__a[..., 0] - e[0], cloud_b__a[..., 1] - e[1], e[2] - cloud_b__a[..., 0],
                                         e[3] - cloud_b__a[..., 1]], dim=-1), dim=-1)[0]
        weights = None
        if self.fov_mode == "ignore_out_fov":
            weights = (min_fov > 0.0).to(torch.float)
        elif self.fov_mode == "use_nearest":
            nearest_dist_sqr_b__a = torch.min(nearest_dist_sqr_b__a, min_fov.square())
        elif self.fov_mode == "mask_close_fov":
            weights = (min_fov > 0.0).to(torch.float) * (nearest_dist_sqr_b__a < min_fov.square())
        loss = self.huber_loss(err_sqr=nearest_dist_sqr_b__a)
        if weights is not None:
            loss = loss * weights
        if self.drop_outliers__perc > 0.0:  # reference :120-133
            keep = 1.0 - self.drop_outliers__perc / 100.0
            bs = loss.size(0)
            kth = int(round(loss.numel() / bs * keep))
            thr = torch.stack([torch.kthvalue(loss[b], kth)[0] for b in range(bs)], dim=0)
            loss = torch.where(loss <= thr[(slice(None),) + tuple([None] * (loss.ndim - 1))], loss, torch.zeros_like(loss))
        return loss


@torch.no_grad()
def get_idx_dists_for_knn(ref_pts, query_pts, num_neighbors: int = 1):
    """reference :138-152"""
    assert query_pts.ndim == 2
    return knn_graph(query_pts, index=ref_pts, k=num_neighbors, loop=True)


def compute_flow_loss_a_to_b(cloud_a, cloud_b, flow_a_to_b, loss_function, nearest_dist_mode: str = "point", knn_indices=None,
                             query_order_indices=None):
    """reference :155-217.  `knn_indices` (optional list of KnnIndex, one per batch row of cloud_b) lets the caller reuse
    the device index of a reference cloud across RAFT iterations / flow types.  `query_order_indices` (optional list of
    KnnIndex of the rows of cloud_a): the queries cloud_a + flow are answered in that cloud's bucket order -- neighbouring
    queries then walk the same cells of the reference index (cache hits instead of scattered reads); the result is the
    same, the search is exact."""
    assert nearest_dist_mode == "point"
    assert cloud_a.ndim == 3 and cloud_b.ndim == 3 and flow_a_to_b.ndim == 3
    cloud_b__a = cloud_a + flow_a_to_b
    bs = cloud_b.size(0)
    if knn_indices is not None:
        # rows that query the SAME reference index (flow types / RAFT iterations stacked along the batch axis) go through
        # one launch: the queries are independent
        groups = {}
        for b in range(bs):
            groups.setdefault(id(knn_indices[b]), []).append(b)
        idx = torch.empty((bs, cloud_b__a.shape[1], 1), dtype=torch.int64, device=cloud_b__a.device)
        for rows in groups.values():
            contiguous = rows == list(range(rows[0], rows[-1] + 1))
            if contiguous:  # plain slices: no index tensor (a host->device copy, which a hipGraph capture refuses)
                q = cloud_b__a[rows[0]:rows[-1] + 1].reshape(-1, cloud_b__a.shape[-1])
            else:
                q = torch.cat([cloud_b__a[b] for b in rows], dim=0)
            perm = None
            if contiguous and query_order_indices is not None and len({id(query_order_indices[b]) for b in rows}) == 1:
                sorted_ids = getattr(query_order_indices[rows[0]], "sorted_ids", None)  # device KnnIndex only
                perm = sorted_ids() if sorted_ids is not None else None
                if perm is not None and perm.shape[0] != cloud_b__a.shape[1]:
                    perm = None
            if perm is not None:
                q = cloud_b__a[rows[0]:rows[-1] + 1].detach().index_select(1, perm).reshape(-1, cloud_b__a.shape[-1])
                res = get_idx_dists_for_knn(knn_indices[rows[0]], q, 1).view(len(rows), -1)
                idx[rows[0]:rows[-1] + 1, :, 0].index_copy_(1, perm, res)
                continue
            res = get_idx_dists_for_knn(knn_indices[rows[0]], q, 1).view(len(rows), -1, 1)
            if contiguous:
                idx[rows[0]:rows[-1] + 1] = res
            else:
                for j, b in enumerate(rows):
                    idx[b] = res[j]
    else:
        idx = torch.stack([get_idx_dists_for_knn(cloud_b[b], cloud_b__a[b], 1) for b in range(bs)], dim=0)
    if (cloud_a.is_cuda and isinstance(loss_function, NearestPointLoss) and loss_function.drop_outliers__perc == 0.0
            and loss_function.fov_mode in NearestPointLoss.FOV_MODES):
        e = [float(v) for v in loss_function.bev_extent]
        cfg = L.NpLossCfg(bs, cloud_a.shape[1], cloud_b.shape[1], (ctypes.c_float * 4)(*e),
                          NearestPointLoss.FOV_MODES[loss_function.fov_mode], float(loss_function.L1_delta))
        loss, d2 = _NearestPointLossFused.apply(cloud_a, flow_a_to_b, cloud_b, idx, cfg)
        return loss, Munch(nearest_dist_sqr=d2, nearest_dist=d2.sqrt())
    nearest = torch.gather(cloud_b, 1, idx.repeat(1, 1, 3))
    d2 = squared_sum(nearest - cloud_b__a, dim=-1)
    loss = loss_function(cloud_b__a=cloud_b__a, nearest_cloud_b__a=nearest, nearest_dist_sqr_b__a=d2)
    return loss, Munch(nearest_dist_sqr=d2, nearest_dist=d2.sqrt())
