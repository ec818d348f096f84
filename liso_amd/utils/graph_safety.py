"""hipGraph replay safety on this stack (ROCm 7.2, MI355X).

Measured root cause of the "graph + eager launches" GPU memory faults / silently wrong replays of rounds 1-2
(scripts/debug_graph_memcpy.py, scripts/debug_pillar_graph_fault.py):

  * a captured **hipMemsetAsync node** stops doing its job after roughly ten thousand kernel launches have been issued on the
    process's streams since the capture (eager launches between replays count; a stream synchronisation before every replay
    does not help): the destination is no longer cleared, something else gets written.  Kernel nodes and memcpy nodes are
    not affected.  With `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` in the environment (the runtime then enqueues the nodes at launch
    time instead of replaying AQL packets recorded at instantiation) the same graphs stay correct, at a higher launch cost
    (LISO loop: 7.8 instead of 6.7 ms per step).
  * memset nodes come from: hipMemsetAsync in the own library (removed: liso_amd/csrc/zero_fill.h is a kernel); ATen
    reductions that need several blocks per output (their inter-block semaphores are zeroed with cudaMemsetAsync); rocPRIM
    device scans / radix sorts behind torch.cumsum / torch.sort (look-back state and histograms); MIOpen / rocSOLVER calls.
    Victims seen: counters not zero -> scattered writes leave their buffer (memory fault: KnnIndex build, the pillar
    voxeliser at B = 4, torch.sort > 1 M keys); reductions returning garbage (SLIM training graph: gradients of 1e26 after 4
    replays with 1500 eager launches in between).

Rules the captured regions of this package follow:
  1. nothing in liso_amd/csrc calls hipMemsetAsync;
  2. global reductions inside a captured region go through `two_stage_amax / two_stage_amin / two_stage_sum` below (every stage
     is reduced by one block per output: no semaphores, no memset);
  3. device sorts / scans whose input depends on the sweeps only are computed eagerly and copied in (BevGatherPlan, the
     dynamicness threshold; until round 4 also the pillar encoder's rocPRIM radix sort -- the voxeliser no longer sorts or calls any
     library, csrc/pillars.hip);
  4. a region that cannot follow 1-3 (the SLIM *training* graph: autograd's reductions) is only captured when
     DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is set: `require_node_replay()` raises otherwise.
"""
import os

import torch

ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def packet_capture_disabled():
    return os.environ.get(ENV, "") == "0"


def require_node_replay(what):
    if not packet_capture_disabled():
        raise RuntimeError(
            f"{what}: the captured region contains hipMemsetAsync nodes (ATen reductions / rocPRIM scans), which the ROCm 7.2 "
            f"runtime stops executing correctly after ~10k launches when it replays pre-recorded packets.  Set {ENV}=0 in the "
            "environment before the process initialises HIP (see liso_amd/utils/graph_safety.py), or run this step eagerly.")


def _rows(n):
    for cols in (2048, 1024, 512, 256, 128, 64):
        if n % cols == 0 and n > cols:
            return cols
    return None


def _two_stage(t, op):
    f = t.detach().reshape(-1) if op != "sum" else t.reshape(-1)
    while f.numel() > 2048:
        cols = _rows(f.numel())
        if cols is None:
            break
        f = f.view(-1, cols)
        f = f.amax(dim=1) if op == "amax" else f.amin(dim=1) if op == "amin" else f.sum(dim=1)
    return f.amax() if op == "amax" else f.amin() if op == "amin" else f.sum()


def two_stage_amax(t):
    """max over all elements; no stage needs more than one block per output (no semaphore memset)"""
    return _two_stage(t, "amax")


def two_stage_amin(t):
    return _two_stage(t, "amin")


def two_stage_sum(t):
    """sum over all elements (other summation order than torch.sum: rows of <= 2048 first)"""
    return _two_stage(t, "sum")


def channel_extrema(maps):
    """(amax, amin) over all but the last dimension of [..., C]; device fp32 maps of <= 4 channels whose pixels are regular rows (a channel
    slice of a channels-last tensor): one pass, two launches (include/liso_slim_decode.h: liso_channel_extrema_f32); else staged like
    two_stage_*"""
    C = maps.shape[-1]
    m = maps.detach()
    if m.is_cuda and m.dtype == torch.float32 and 1 <= C <= 4 and m.numel() > 0 and m.stride(-1) == 1:
        rows, st, regular = m.numel() // C, m.stride(-2) if m.dim() > 1 else C, True
        acc = st
        for d in range(m.dim() - 2, -1, -1):  # rows st floats apart throughout
            if m.shape[d] != 1 and m.stride(d) != acc:
                regular = False
                break
            acc *= m.shape[d]
        if regular and st >= C:
            from liso_amd import _lib as L

            lib = L.lib()
            nbytes = lib.liso_channel_extrema_workspace_bytes()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=m.device)
            out = torch.empty(2 * C, dtype=torch.float32, device=m.device)
            with torch.cuda.device(m.device):
                L.check(lib.liso_channel_extrema_f32(L.ptr(m), rows, st, C, L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr()), "channel_extrema")
            return out[:C], out[C:]
    f = maps.detach().reshape(-1, C)
    hi = lo = f
    while hi.shape[0] > 2048:
        cols = _rows(hi.shape[0])
        if cols is None:
            break
        hi = hi.view(-1, cols, C).amax(dim=1)
        lo = lo.view(-1, cols, C).amin(dim=1)
    return hi.amax(dim=0), lo.amin(dim=0)
