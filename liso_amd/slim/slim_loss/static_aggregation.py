"""Static-flow aggregation.  Mirror of liso/slim/slim_loss/static_aggregation.py:8-110."""
import torch

from liso_amd import _lib as L
from liso_amd.slim.slim_loss.weighted_pc_alignment import batched_weighted_pc_alignment, weighted_pc_alignment  # noqa: F401


class BevGatherPlan:
    """The cell of every point of a batch of clouds, flattened over (b, row, col), and the same list sorted by cell.
    Built once per cloud and reused by every BEV->point gather (forward) and its segmented-sum adjoint (backward) of a
    training step: 6 RAFT iterations x 3 gathers per direction."""

    def __init__(self, pointwise_voxel_coordinates_fs, pointwise_valid_mask, grid_hw):
        H, W = int(grid_hw[0]), int(grid_hw[1])
        B, N = pointwise_valid_mask.shape
        c, v = pointwise_voxel_coordinates_fs, pointwise_valid_mask
        if (c.is_cuda and c.dtype in (torch.int32, torch.int64) and v.dtype in (torch.bool, torch.uint8) and c.is_contiguous()
                and v.is_contiguous() and tuple(c.shape) == (B, N, 2)):
            self.lin = torch.empty(B * N, dtype=torch.int32, device=c.device)  # one launch (include/liso_slim.h: liso_bev_lin_index)
            with torch.cuda.device(c.device):
                L.check(L.lib().liso_bev_lin_index(L.ptr(c), int(c.dtype == torch.int64), L.ptr(v), B, N, H, W, L.ptr(self.lin), L.stream_ptr()),
                        "bev_lin_index")
        else:
            c = c.long()
            b = torch.arange(B, device=c.device)[:, None]
            lin = (b * H + c[..., 0]) * W + c[..., 1]
            self.lin = torch.where(pointwise_valid_mask, lin, -1).to(torch.int32).reshape(-1).contiguous()
        self.shape = (B, N, H, W)
        self._sorted = None

    @classmethod
    def tiled(cls, pointwise_voxel_coordinates_fs, pointwise_valid_mask, grid_hw, n_it, half):
        """The plan of the batch [samples[:half]] * n_it + [samples[half:]] * n_it -- what the decoder sees when the network outputs of
        all RAFT iterations and both flow directions are decoded at once (slim.py: forward) -- from the 2 * half DISTINCT samples: the
        same cloud repeated n_it times has the same cells, so one segmented sort of the distinct clouds (2 x 120k keys instead of
        2.9 M) is expanded with per-copy offsets.  Rows of one cell stay consecutive and in point order inside every copy's block,
        which is all the segmented-sum adjoint asks for (liso_bev_gather_bwd_f32): its sums are bit for bit those of the flat sort."""
        small = cls(pointwise_voxel_coordinates_fs, pointwise_valid_mask, grid_hw)
        n2, N = pointwise_valid_mask.shape
        H, W = int(grid_hw[0]), int(grid_hw[1])
        assert 0 < half <= n2 and n_it >= 1
        dev = small.lin.device
        if small.lin.is_cuda:  # one launch (include/liso_slim.h: liso_bev_plan_tile_lin); the sorted view: _sort, two more
            assert n2 * H * W * n_it < 2 ** 31
            self = object.__new__(cls)
            self.lin = torch.empty(n2 * n_it * N, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                L.check(L.lib().liso_bev_plan_tile_lin(L.ptr(small.lin), n2, N, n_it, half, H, W, L.ptr(self.lin), L.stream_ptr()),
                        "bev_plan_tile_lin")
            self.shape = (n2 * n_it, N, H, W)
            self._sorted = None
            self._tile_dev = (small.lin, n2, N, n_it, half, H, W)
            return self
        src = torch.cat([torch.arange(half, device=dev).repeat(n_it), torch.arange(half, n2, device=dev).repeat(n_it)])
        off = ((torch.arange(src.numel(), device=dev) - src) * (H * W)).to(torch.int32)[:, None]
        rows = small.lin.view(n2, N)
        self = object.__new__(cls)
        picked = rows[src]
        self.lin = torch.where(picked >= 0, picked + off, picked).reshape(-1).contiguous()
        self.shape = (int(src.numel()), N, H, W)
        self._sorted = None
        self._tile = (rows, src, off, N)
        return self

    def _sort(self):
        """the cell-sorted view of the list: only the adjoint needs it (inference never builds it)"""
        if self._sorted is None and getattr(self, "_tile_dev", None) is not None:
            # device tensors: ONE flat (radix) sort of the distinct samples' keys + two launches (rank inside the runs; expansion to the
            # tiled batch with per-copy offsets) -- include/liso_slim.h: liso_bev_plan_rank / liso_bev_plan_expand
            rows, n2, N, n_it, half, H, W = self._tile_dev
            dev = rows.device
            s_flat, o_flat = torch.sort(rows, stable=True)
            rank_flat = torch.empty_like(s_flat)
            total = n2 * n_it * N
            sorted_lin, order, rank = (torch.empty(total, dtype=torch.int32, device=dev) for _ in range(3))
            with torch.cuda.device(dev):
                lib = L.lib()
                L.check(lib.liso_bev_plan_rank(L.ptr(s_flat), None, s_flat.numel(), L.ptr(rank_flat), None, L.stream_ptr()), "bev_plan_rank")
                L.check(lib.liso_bev_plan_expand(L.ptr(s_flat), L.ptr(o_flat), L.ptr(rank_flat), n2, N, n_it, half, H, W, L.ptr(sorted_lin),
                                                 L.ptr(order), L.ptr(rank), L.stream_ptr()), "bev_plan_expand")
            self._sorted = (sorted_lin, order, rank)
        if self._sorted is None and getattr(self, "_tile", None) is not None:
            rows, src, off, N = self._tile
            n2 = rows.shape[0]
            dev = rows.device
            # ONE flat (radix) sort of the distinct samples' keys (a per-row sort of two long rows takes torch's merge-sort path:
            # 24 launches, 0.14 ms).  The flat order is [invalid rows of every sample | sample 0's cells | sample 1's cells | ...]:
            # sample b's valid rows are the range [start_b, start_b + nv_b) of it; every copy's block is those rows followed by
            # invalid padding up to N rows (the adjoint skips cell < 0), all index arithmetic on the device.
            s_flat, o_flat = torch.sort(rows.reshape(-1), stable=True)
            pos = torch.arange(s_flat.numel(), device=dev, dtype=torch.int32)
            rank_flat = pos - torch.searchsorted(s_flat, s_flat, right=False).to(torch.int32)
            nv = (rows >= 0).sum(dim=1)                                   # valid rows per distinct sample
            start = (rows.numel() - nv.sum()) + torch.cumsum(nv, 0) - nv  # first valid row of sample b in the flat order
            k = torch.arange(N, device=dev)[None, :]
            idx = (start[src][:, None] + k).clamp(max=s_flat.numel() - 1)
            ok = k < nv[src][:, None]
            sorted_lin = torch.where(ok, s_flat[idx] + off, torch.full_like(off, -1)).reshape(-1).contiguous()
            local = (o_flat[idx] - (src * N)[:, None]).to(torch.int32)   # row inside its sample
            order = torch.where(ok, local + (torch.arange(src.numel(), device=dev, dtype=torch.int32) * N)[:, None], torch.zeros_like(local))
            rank = torch.where(ok, rank_flat[idx], torch.zeros_like(local))
            self._sorted = (sorted_lin, order.reshape(-1).contiguous(), rank.reshape(-1).contiguous())
        if self._sorted is None and self.lin.is_cuda:
            sorted_lin, order64 = torch.sort(self.lin, stable=True)
            seg_rank, order = torch.empty_like(sorted_lin), torch.empty_like(sorted_lin)
            with torch.cuda.device(self.lin.device):
                L.check(L.lib().liso_bev_plan_rank(L.ptr(sorted_lin), L.ptr(order64), sorted_lin.numel(), L.ptr(seg_rank), L.ptr(order),
                                                   L.stream_ptr()), "bev_plan_rank")
            self._sorted = (sorted_lin, order, seg_rank)
        if self._sorted is None:
            sorted_lin, order = torch.sort(self.lin, stable=True)
            pos = torch.arange(sorted_lin.numel(), device=self.lin.device, dtype=torch.int32)
            # first row of every run of equal cells = lower bound of the value in the sorted list itself
            seg_rank = (pos - torch.searchsorted(sorted_lin, sorted_lin, right=False).to(torch.int32)).contiguous()
            self._sorted = (sorted_lin.contiguous(), order.to(torch.int32).contiguous(), seg_rank)
        return self._sorted

    def prepare_backward(self):
        """build the sorted view now (callers that capture the backward pass into a hipGraph: the sort must stay outside)"""
        self._sort()
        return self

    @property
    def sorted_lin(self):
        return self._sort()[0]

    @property
    def order(self):
        return self._sort()[1]

    @property
    def seg_rank(self):
        return self._sort()[2]

    @property
    def lin64(self):
        """`lin` as int64 (torch indexing), built on first use"""
        if getattr(self, "_lin64", None) is None:
            self._lin64 = self.lin.long()
        return self._lin64

    def matches(self, grid_data, mask):
        B, N, H, W = self.shape
        return tuple(grid_data.shape[:3]) == (B, H, W) and tuple(mask.shape) == (B, N)


class _BevGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid_data, plan, default_value):
        L.require_cuda(grid_data)
        B, N, H, W = plan.shape
        C = grid_data.shape[-1]
        g = grid_data.float().contiguous()
        out = torch.empty((B, N, C), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.check(L.lib().liso_bev_gather_fwd_f32(L.ptr(g), L.ptr(plan.lin), B * N, C, float(default_value), L.ptr(out),
                                                    L.stream_ptr()), "bev_gather_fwd")
        ctx.plan, ctx.C = plan, C
        return out

    @staticmethod
    def backward(ctx, grad_out):
        plan, C = ctx.plan, ctx.C
        B, N, H, W = plan.shape
        go = grad_out.float().contiguous()
        gg = torch.zeros((B, H, W, C), dtype=torch.float32, device=go.device)
        partial = torch.empty((B * N, C), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            L.check(L.lib().liso_bev_gather_bwd_f32(L.ptr(go), L.ptr(plan.sorted_lin), L.ptr(plan.order), L.ptr(plan.seg_rank),
                                                    B * N, C, L.ptr(partial), L.ptr(gg), L.stream_ptr()), "bev_gather_bwd")
        return gg, None, None


def batched_grid_data_to_pointwise_data(grid_data, pointwise_voxel_coordinates_fs, pointwise_valid_mask, default_value, plan=None):
    """reference :8-31 -- gather [B,H,W,C] at each point's pillar (invalid rows -> default), out of place.
    Float maps go through the gfx950 gather of include/liso_slim.h (`plan`: reusable BevGatherPlan); other dtypes use
    torch indexing (they carry no gradient)."""
    assert len(grid_data.shape) == 4, grid_data.shape
    if grid_data.dtype == torch.float32:
        if plan is None or not plan.matches(grid_data, pointwise_valid_mask):
            plan = BevGatherPlan(pointwise_voxel_coordinates_fs, pointwise_valid_mask, grid_data.shape[1:3])
        return _BevGather.apply(grid_data, plan, default_value)
    coors = torch.where(pointwise_valid_mask[..., None], pointwise_voxel_coordinates_fs,
                        torch.zeros_like(pointwise_voxel_coordinates_fs)).long()
    b = torch.arange(pointwise_valid_mask.shape[0], device=grid_data.device)[:, None].expand(-1, pointwise_valid_mask.shape[1])
    data = grid_data[b, coors[..., 0], coors[..., 1]]
    return torch.where(pointwise_valid_mask[..., None], data, default_value)


def compute_batched_bev_static_aggregated_flow(pc, pointwise_voxel_coordinates_fs, pointwise_valid_mask, static_flow_bev,
                                               staticness_weights, voxel_center_metric_coordinates_bev,
                                               use_eps_for_weighted_pc_alignment: bool = False, plan=None):
    """reference :34-110 -- per sample: Kabsch of (points, points + static flow) weighted by staticness, then the rigid
    flow field (T - I) applied to every BEV cell centre.  Invalid (padding) points enter the weighted moments with
    weight 0 instead of being removed by boolean indexing (same sums, no device->host sync)."""
    assert len(static_flow_bev.shape) == 4 and static_flow_bev.shape[-1] == 2
    both = torch.cat([static_flow_bev, staticness_weights[..., None]], dim=-1)  # one gather for flow + weight
    pw = batched_grid_data_to_pointwise_data(both, pointwise_voxel_coordinates_fs, pointwise_valid_mask, 0.0, plan=plan)
    pw_flow = torch.cat([pw[..., :2], torch.zeros_like(pw[..., :1])], dim=-1)
    pw_static = pw[..., 2]
    centers = voxel_center_metric_coordinates_bev
    # all samples of the batch in one set of launches (the reference loops over the batch, :60-106)
    p0 = torch.where(pointwise_valid_mask[..., None], pc[..., :3], 0.0)
    T, nep = batched_weighted_pc_alignment(p0, p0 + pw_flow, pw_static, pointwise_valid_mask,
                                           use_epsilon_on_weights=use_eps_for_weighted_pc_alignment)
    # (T - I) applied to every cell centre (z = 0, w = 1) as fp64 broadcast multiply-adds; the reference's einsum
    # (:88-99) is a [4x4]x[4xHW] DGEMM that rocBLAS runs with a 128x128 tile: 28 ms per call at 512^2 (measured)
    D = T - torch.eye(4, dtype=torch.float64, device=T.device)
    flows = (D[:, None, None, :2, 0] * centers[None, ..., 0:1] + D[:, None, None, :2, 1] * centers[None, ..., 1:2]
             + D[:, None, None, :2, 3]).float()
    return flows, T, nep


def compute_pointwise_static_aggregated_flow(pc, pointwise_valid_mask, static_flow_pt, staticness_weights_pt, cell_centers_pt,
                                             use_eps_for_weighted_pc_alignment: bool = False):
    """compute_batched_bev_static_aggregated_flow for quantities that are already per point: static_flow_pt [B,N,2] and
    staticness_weights_pt [B,N] are the BEV maps of the reference (:34-110) read at every point's pillar (0 at padding
    rows), cell_centers_pt [B,N,2] (fp64) the metric centre of that pillar.  Returns the rigid flow (T - I) of the pillar
    centre per point [B,N,2] -- the value the reference's BEV `static_aggr_flow` map holds at the point's pillar -- T, and
    the not-enough-points flags."""
    pw_flow = torch.cat([static_flow_pt, torch.zeros_like(static_flow_pt[..., :1])], dim=-1)
    p0 = torch.where(pointwise_valid_mask[..., None], pc[..., :3], 0.0)
    T, nep = batched_weighted_pc_alignment(p0, p0 + pw_flow, staticness_weights_pt, pointwise_valid_mask,
                                           use_epsilon_on_weights=use_eps_for_weighted_pc_alignment)
    D = T - torch.eye(4, dtype=torch.float64, device=T.device)
    flows = (D[:, None, :2, 0] * cell_centers_pt[..., 0:1] + D[:, None, :2, 1] * cell_centers_pt[..., 1:2] + D[:, None, :2, 3]).float()
    return flows, T, nep
