"""correlation lookup, per-query kernel vs tiled kernel: 4 sweep pairs at 64 x 64 queries, D = 128, 4 levels, radius 3 (the SLIM inference
replay's shape), smooth flow + `--spread` pixels of noise.  hipGraph of 20 launches, median of 5 replays.
    python scripts/corr_lookup_times.py [--spread 0.5] [--batch 4]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd import _lib as L  # noqa: E402
from liso_amd.slim.model.raft_code.utils import coords_grid  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spread", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    B, h, w, D, levels, radius = a.batch, 64, 64, 128, 4, 3
    f1 = torch.randn(B, h * w, D, device="cuda")
    lv = [torch.randn(B, h >> i, w >> i, D, device="cuda") for i in range(levels)]
    base = coords_grid(B, h, w, device="cuda")
    coords = (base + 1.7 + 0.01 * base + a.spread * torch.randn(B, 2, h, w, device="cuda")).contiguous()
    cfg = L.CorrCfg(B, h, w, D, levels, radius)
    ptrs = (ctypes.c_void_p * levels)(*[t.data_ptr() for t in lv])
    out = torch.empty((B, h, w, levels * 49), device="cuda")
    for name, fn in (("per-query", L.lib().liso_corr_lookup_fwd_f32), ("tiled", L.lib().liso_corr_lookup_fwd_tiled_f32)):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            L.check(fn(ctypes.byref(cfg), L.ptr(f1), ptrs, L.ptr(coords), L.ptr(out), ctypes.c_void_p(s.cuda_stream)), name)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                L.check(fn(ctypes.byref(cfg), L.ptr(f1), ptrs, L.ptr(coords), L.ptr(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), name)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{name:10s} batch {B} spread {a.spread}: {sorted(ts)[2]:.1f} us per lookup")


if __name__ == "__main__":
    main()
