"""liso_corr_bwd_features_f32 (the two contractions of the correlation backward, all levels per launch) against the eight batched
library GEMMs it replaced, at the SLIM training step's shape (B = 2 flow directions, 64 x 64 queries, D = 128) and at 1024^2's.
python scripts/corr_bwd_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.slim.model.raft_code.corr import corr_bwd_features  # noqa: E402
from liso_amd.utils import mfma_conv as MC  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return 1e3 * t[len(t) // 2]


for B, h, w, D in ((2, 64, 64, 128), (1, 64, 64, 128), (2, 128, 128, 128)):
    g = torch.Generator().manual_seed(0)
    hw = h * w
    f1 = torch.randn(B, hw, D, generator=g).cuda()
    levels = [torch.randn(B, h >> i, w >> i, D, generator=g).cuda() for i in range(4)]
    dvol = [torch.randn(B, hw, l.shape[1] * l.shape[2], generator=g).cuda() for l in levels]
    flop = 2 * 2.0 * B * hw * D * sum(dv.shape[2] for dv in dvol)

    def lib():
        g1 = None
        for dv, l in zip(dvol, levels):
            f2m = l.reshape(B, -1, D)
            g1 = torch.bmm(dv, f2m) if g1 is None else torch.baddbmm(g1, dv, f2m)
            torch.bmm(dv.transpose(1, 2), f1)

    row = [f"B={B} {h}x{w} D={D}: {flop / 1e9:.1f} GFLOP"]
    for mode in ("x3", "exact"):
        prev = MC.set_fp32_mode(mode)
        t = timed(lambda: corr_bwd_features(f1, levels, dvol))
        MC.set_fp32_mode(prev)
        row.append(f"own[{mode}] {t:7.1f} us = {flop / t / 1e6:6.1f} TFLOP/s")
    t = timed(lib)
    row.append(f"rocBLAS fp32 (8 launches) {t:7.1f} us = {flop / t / 1e6:6.1f} TFLOP/s")
    print("   ".join(row), flush=True)
