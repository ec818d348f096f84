"""Generates tests/golden/track_smoothing_reference.npz from the reference's own liso/tracker/track_smoothing.py:
  smooth_track_jerk (:104-290) for max_iters in {1, 3, 20, 200, 2000} on padded batches of tracks, with return_losses=True.
The optimisation is NOT well conditioned at its default settings (Adam, lr 0.1, no decay): a 1e-6 change of one input coordinate
moves the 2000-iteration result by 4e-2 m (measured with the reference itself, stored below as `*_sensitivity`), so the fixture
holds short runs (tight comparison) and the full run (comparison within the reference's own sensitivity, and of the final loss).
Absent third-party modules are stubbed with empty modules.  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_track_smoothing_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def tracks(g, B, T, lengths):
    t = np.arange(T)[None, :, None].astype(np.float64)
    speed, curve = g.uniform(0.5, 1.5, (B, 1, 1)), g.uniform(-0.02, 0.02, (B, 1, 1))
    heading = g.uniform(-3, 3, (B, 1, 1))
    s = speed * t + curve * t ** 2
    pos = np.concatenate([g.uniform(-20, 20, (B, 1, 1)) + s * np.cos(heading), g.uniform(-20, 20, (B, 1, 1)) + s * np.sin(heading),
                          np.full((B, T, 1), -0.8)], -1) + g.normal(0, 0.25, (B, T, 3)) * [1, 1, 0.2]
    valid = np.arange(T)[None, :] < np.array(lengths)[:, None]
    pos[~valid] = 0.0
    yaw = heading.repeat(T, 1) + g.normal(0, 0.1, (B, T, 1))
    return pos.astype(np.float32), valid, yaw.astype(np.float32)


def main():
    def _imp():
        import liso.tracker.track_smoothing as ts
        return ts

    ts = import_with_stubs(_imp)
    g = np.random.default_rng(31)
    out = {}
    for tag, (B, T, lengths) in {"a": (3, 24, [24, 18, 10]), "b": (1, 60, [60]), "c": (2, 4, [4, 4])}.items():
        pos, valid, yaw = tracks(g, B, T, lengths)
        # (smooth_track_jerk writes the aligned headings INTO its yaw argument -- `.detach()` shares the storage, :230 -- and returns
        # that tensor: every call gets its own copy and every result is copied out)
        out[f"{tag}_pos"], out[f"{tag}_valid"], out[f"{tag}_yaw"] = pos.copy(), valid.copy(), yaw.copy()
        for iters in ((1, 3, 20, 200, 2000) if T > 4 else (20,)):
            res = ts.smooth_track_jerk(torch.from_numpy(pos.copy()), torch.from_numpy(valid.copy()), torch.from_numpy(yaw.copy()), 0.1,
                                       max_iters=iters, return_losses=T > 4)
            out[f"{tag}_{iters}_pos"], out[f"{tag}_{iters}_rot"], out[f"{tag}_{iters}_velo"] = (r.numpy().copy() for r in res[:3])
            if T > 4:
                out[f"{tag}_{iters}_last_loss"] = res[3][-1]["per_batch_loss"]
                out[f"{tag}_{iters}_last_jerk_loss"] = res[3][-1]["per_batch_jerk_loss"]
        if T > 4:
            p2 = pos.copy()
            p2[0, 5, 0] += 1e-6
            res2 = ts.smooth_track_jerk(torch.from_numpy(p2), torch.from_numpy(valid.copy()), torch.from_numpy(yaw.copy()), 0.1)
            out[f"{tag}_sensitivity"] = np.array(float((res2[0] - torch.from_numpy(out[f"{tag}_2000_pos"])).abs().max()))
            print(tag, "sensitivity of the reference to a 1e-6 input change:", float(out[f"{tag}_sensitivity"]))
    np.savez_compressed(os.path.join(HERE, "track_smoothing_reference.npz"), **out)
    print("arrays:", len(out))


if __name__ == "__main__":
    main()
