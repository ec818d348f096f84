"""knn_graph: exact 1-nearest-neighbour indices.  Mirror of liso/slim/slim_loss/knn_graph.py:10-98 for the call shape
the SLIM loss uses (`knn_graph(x, index=ref, k=1, loop=True)` -> LongTensor[Nq, 1], knn_wrapper.py:139-152).

The reference copies both clouds to the host and builds a pynanoflann KD-tree per call; here the reference cloud is
bucketed on the device once (`KnnIndex`) and queried by the exact grid-ring kernel of include/liso_slim.h.  Ties are
resolved towards the smaller index (pynanoflann's tie order is unspecified; the loss only consumes distances).
"""
import ctypes

import torch

from liso_amd import _lib as L


class _Grid:
    def __init__(self, ref, lo, hi, cell, max_cells_per_side, z_min, z_cell, nz):
        ext = max(hi[0] - lo[0], hi[1] - lo[1], 1e-3)
        cell = max(cell, ext / max_cells_per_side)
        nx = max(1, int((hi[0] - lo[0]) / cell) + 1)
        ny = max(1, int((hi[1] - lo[1]) / cell) + 1)
        self.grid = L.KnnGrid(lo[0], lo[1], cell, nx, ny, z_min, z_cell, nz)
        lib = L.lib()
        n = ref.shape[0]
        nbytes = lib.liso_knn_workspace_bytes(ctypes.byref(self.grid), n)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
        L.check(lib.liso_knn_build_f32(ctypes.byref(self.grid), L.ptr(ref), ref.shape[1], n, L.ptr(self.ws), nbytes,
                                       L.stream_ptr()), "knn_build")


class KnnIndex:
    """Device-resident two-level uniform-grid index over one reference cloud; reusable across queries.
    fine grid (0.2 m cells): dense near field resolved in 1-2 rings; coarse grid (2 m cells): the few queries that land in
    empty space.  Both passes are exact; the coarse pass only touches rows the fine pass could not prove.  Inside every
    xy cell the points are ordered by z bin, so a query next to a wall reads only the slice of the stack it can reach."""

    FINE_RINGS = 6

    def __init__(self, ref: torch.Tensor, cell: float = 0.2, coarse_cell: float = 2.0, extent=None, all_rows_finite=False):
        """`all_rows_finite`: the caller knows that no row of `ref` holds NaN / inf (enables `sorted_ids`)"""
        assert ref.ndim == 2 and ref.shape[1] >= 3
        L.require_cuda(ref)
        self.all_rows_finite, self._sorted_ids = bool(all_rows_finite), None
        self.ref = ref.detach().float().contiguous()
        n = self.ref.shape[0]
        if extent is not None:  # (x_min, y_min, x_max, y_max) known up front (the BEV range): no host sync at all;
            lo, hi = [float(extent[0]), float(extent[1])], [float(extent[2]), float(extent[3])]  # outliers are clamped
        elif n > 0:  # one small sync per index build; rows marked NaN (padding, knn_loss.py:44-45) do not count
            xy = self.ref[:, :2]
            fin = torch.isfinite(xy)
            lo = torch.where(fin, xy, float("inf")).amin(dim=0).tolist()
            hi = torch.where(fin, xy, float("-inf")).amax(dim=0).tolist()
            if not all(v == v and abs(v) != float("inf") for v in lo + hi):
                lo, hi = [0.0, 0.0], [1.0, 1.0]
        else:
            lo, hi = [0.0, 0.0], [1.0, 1.0]
        with torch.cuda.device(ref.device):
            # z bins over [-4 m, 4 m) (clamped outside): 0.25 m inside the fine cells, 2 m inside the coarse ones
            self.fine = _Grid(self.ref, lo, hi, cell, 700, -4.0, 0.25, 32)
            self.coarse = _Grid(self.ref, lo, hi, coarse_cell, 700, -4.0, 2.0, 4)

    def sorted_ids(self):
        """int64 [n]: this cloud's rows in (fine) bucket order -- a spatially coherent visiting order for queries that start
        at this cloud's points; None unless the index was built with all_rows_finite=True"""
        if not self.all_rows_finite:
            return None
        if self._sorted_ids is None:
            n = self.ref.shape[0]
            ids = torch.empty(n, dtype=torch.int64, device=self.ref.device)
            with torch.cuda.device(self.ref.device):
                L.check(L.lib().liso_knn_sorted_ids(ctypes.byref(self.fine.grid), L.ptr(self.fine.ws), n, L.ptr(ids), L.stream_ptr()),
                        "knn_sorted_ids")
            self._sorted_ids = ids
        return self._sorted_ids

    def query(self, x: torch.Tensor, return_dist_sqr=False):
        q = x.detach().float().contiguous()
        nq = q.shape[0]
        idx = torch.empty(nq, dtype=torch.int64, device=q.device)
        d2 = torch.empty(nq, dtype=torch.float32, device=q.device) if return_dist_sqr else None
        lib = L.lib()
        n = self.ref.shape[0]

        with torch.cuda.device(q.device):
            L.check(L.TIMER.launch("knn_query", lambda: lib.liso_knn_query_f32(
                ctypes.byref(self.fine.grid), L.ptr(self.fine.ws), ctypes.byref(self.coarse.grid), L.ptr(self.coarse.ws), n,
                L.ptr(q), q.shape[1], nq, L.ptr(idx), L.ptr(d2) if d2 is not None else None, self.FINE_RINGS,
                L.stream_ptr()), units=nq), "knn_query")
        return (idx, d2) if return_dist_sqr else idx


@torch.no_grad()
def knn_graph(x, *, index=None, k: int, batch=None, loop: bool = False, flow: str = "source_to_target", cosine=False,
              num_workers: int = 1, return_kd_tree: bool = False):
    assert flow == "source_to_target" and not cosine and num_workers == 1
    if index is None or k != 1 or not loop or batch is not None or return_kd_tree:
        raise NotImplementedError("only knn_graph(x, index=ref, k=1, loop=True) is on the hot path (knn_wrapper.py:139-152)")
    idx = (index if isinstance(index, KnnIndex) else KnnIndex(index)).query(x)
    return idx[:, None]
