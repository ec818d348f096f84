"""BevGatherPlan.tiled (round 6): the point -> cell plan of the batch the SLIM decoder sees when all RAFT iterations and both flow
directions are decoded at once -- [samples[:B]] * n_it + [samples[B:]] * n_it -- built from ONE sort of the distinct samples and
expanded with per-copy offsets, against the flat construction over the tiled batch (host tensors: the plan's torch path)."""
import pytest
import torch


def _runs(sorted_lin, order):
    d = {}
    for c, o in zip(sorted_lin.tolist(), order.tolist()):
        if c >= 0:
            d.setdefault(c, []).append(o)
    return d


@pytest.mark.parametrize("n2,N,H,W,n_it,half,p_valid", [(4, 500, 16, 16, 3, 2, 0.8), (2, 300, 8, 8, 6, 1, 0.5), (2, 64, 4, 4, 2, 1, 1.0),
                                                         (2, 40, 4, 4, 3, 1, 0.0)])
def test_tiled_plan_equals_the_flat_plan_of_the_tiled_batch(n2, N, H, W, n_it, half, p_valid):
    from liso_amd.slim.slim_loss.static_aggregation import BevGatherPlan

    g = torch.Generator().manual_seed(N)
    coors = torch.randint(0, H, (n2, N, 2), generator=g)
    valid = torch.rand(n2, N, generator=g) < p_valid
    tile = lambda t: torch.cat([t[:half]] * n_it + [t[half:]] * n_it, 0)  # noqa: E731
    flat = BevGatherPlan(tile(coors), tile(valid), (H, W))
    til = BevGatherPlan.tiled(coors, valid, (H, W), n_it, half)
    assert til.shape == flat.shape and torch.equal(flat.lin, til.lin)
    # the adjoint's contract: rows of one cell consecutive, in point order, with their rank in the run; same row count and dtypes
    assert _runs(flat.sorted_lin, flat.order) == _runs(til.sorted_lin, til.order)
    sl, rk = til.sorted_lin.tolist(), til.seg_rank.tolist()
    i = 0
    while i < len(sl):
        j = i
        while j < len(sl) and sl[j] == sl[i]:
            j += 1
        if sl[i] >= 0:
            assert rk[i:j] == list(range(j - i)), (i, rk[i:j])
        i = j
    for a, b in ((til.sorted_lin, flat.sorted_lin), (til.order, flat.order), (til.seg_rank, flat.seg_rank)):
        assert a.shape == b.shape and a.dtype == b.dtype == torch.int32 and a.is_contiguous()
