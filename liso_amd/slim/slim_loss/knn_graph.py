"""knn_graph: exact 1-nearest-neighbour indices.  Mirror of liso/slim/slim_loss/knn_graph.py:10-98 for the call shape
the SLIM loss uses (`knn_graph(x, index=ref, k=1, loop=True)` -> LongTensor[Nq, 1], knn_wrapper.py:139-152).

The reference copies both clouds to the host and builds a pynanoflann KD-tree per call; here the reference cloud is
bucketed on the device once (`KnnIndex`) and queried by the exact grid-ring kernel of include/liso_slim.h.  Ties are
resolved towards the smaller index (pynanoflann's tie order is unspecified; the loss only consumes distances).
"""
import ctypes

import torch

from liso_amd import _lib as L


class KnnIndex:
    """Device-resident uniform-grid index over one reference cloud; reusable across queries."""

    def __init__(self, ref: torch.Tensor, cell: float = 0.5, max_cells_per_side: int = 1000):
        assert ref.ndim == 2 and ref.shape[1] >= 3
        L.require_cuda(ref)
        self.ref = ref.detach().float().contiguous()
        n = self.ref.shape[0]
        if n > 0:
            lo = self.ref[:, :2].amin(dim=0)
            hi = self.ref[:, :2].amax(dim=0)
            lo_h, hi_h = lo.tolist(), hi.tolist()  # one small sync per index build (twice per training step)
        else:
            lo_h, hi_h = [0.0, 0.0], [1.0, 1.0]
        ext = max(hi_h[0] - lo_h[0], hi_h[1] - lo_h[1], 1e-3)
        cell = max(cell, ext / max_cells_per_side)
        nx = max(1, int((hi_h[0] - lo_h[0]) / cell) + 1)
        ny = max(1, int((hi_h[1] - lo_h[1]) / cell) + 1)
        self.grid = L.KnnGrid(lo_h[0], lo_h[1], cell, nx, ny)
        lib = L.lib()
        nbytes = lib.liso_knn_workspace_bytes(ctypes.byref(self.grid), n)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
        with torch.cuda.device(ref.device):
            L.check(lib.liso_knn_build_f32(ctypes.byref(self.grid), L.ptr(self.ref), self.ref.shape[1], n, L.ptr(self.ws),
                                           nbytes, L.stream_ptr()), "knn_build")

    def query(self, x: torch.Tensor, return_dist_sqr=False):
        q = x.detach().float().contiguous()
        nq = q.shape[0]
        idx = torch.empty(nq, dtype=torch.int64, device=q.device)
        d2 = torch.empty(nq, dtype=torch.float32, device=q.device) if return_dist_sqr else None
        with torch.cuda.device(q.device):
            L.check(L.TIMER.launch("knn_query", lambda: L.lib().liso_knn_query_f32(
                ctypes.byref(self.grid), L.ptr(self.ref), self.ref.shape[1], self.ref.shape[0], L.ptr(self.ws), L.ptr(q),
                q.shape[1], nq, L.ptr(idx), L.ptr(d2) if d2 is not None else None, L.stream_ptr())), "knn_query")
        return (idx, d2) if return_dist_sqr else idx


@torch.no_grad()
def knn_graph(x, *, index=None, k: int, batch=None, loop: bool = False, flow: str = "source_to_target", cosine=False,
              num_workers: int = 1, return_kd_tree: bool = False):
    assert flow == "source_to_target" and not cosine and num_workers == 1
    if index is None or k != 1 or not loop or batch is not None or return_kd_tree:
        raise NotImplementedError("only knn_graph(x, index=ref, k=1, loop=True) is on the hot path (knn_wrapper.py:139-152)")
    idx = (index if isinstance(index, KnnIndex) else KnnIndex(index)).query(x)
    return idx[:, None]
