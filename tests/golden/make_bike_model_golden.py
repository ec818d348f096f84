"""Generates tests/golden/bike_model_reference.npz from the reference's own liso/tracker/track_smoothing.py:
  BatchedBikeModel.forward (forward_compiled / car_dynamics, :300-337,490-528) on given inputs and parameters, with the gradient of
  a fixed function of the states with respect to every parameter;
  smooth_track_bike_model (:577-741) for max_iters in {1, 3, 30} with return_losses=True, and the movement of its result when one
  input coordinate changes by 1e-6 (the conditioning of L-BFGS with strong-Wolfe line search on this loss).
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_bike_model_golden.py"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402
from make_track_smoothing_golden import tracks  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def main():
    def _imp():
        import liso.tracker.track_smoothing as ts
        return ts

    ts = import_with_stubs(_imp)
    g = np.random.default_rng(77)
    out = {}
    for tag, (B, T, lengths) in {"a": (3, 24, [24, 18, 10]), "b": (5, 40, [40, 33, 40, 12, 25])}.items():
        pos, valid, yaw = tracks(g, B, T, lengths)
        length = g.uniform(3.5, 5.0, B).astype(np.float32)
        out[f"{tag}_pos"], out[f"{tag}_valid"], out[f"{tag}_yaw"], out[f"{tag}_length"] = pos, valid, yaw, length
        # ---- the rollout alone ----------------------------------------------------------------------------------------------------
        m = ts.BatchedBikeModel(batched_observed_track_pos=torch.from_numpy(pos.copy()), batched_vehicle_length=torch.from_numpy(length),
                                time_between_frames_s=0.1, max_yaw_rate=np.pi / 2, max_velocity=50.0)
        with torch.no_grad():
            m.accel_over_time.copy_(torch.from_numpy(g.normal(0, 3.0, (B, T)).astype(np.float32)))
            m.steering_input_over_time.copy_(torch.from_numpy(g.normal(0, 2.0, (B, T)).astype(np.float32)))
        w = torch.from_numpy(g.normal(0, 1.0, (B, T, 5)).astype(np.float32))
        states = m.forward()
        (states * w).sum().backward()
        out[f"{tag}_rollout_w"] = w.numpy()
        out[f"{tag}_rollout_states"] = states.detach().numpy()
        for name, p in m.named_parameters():
            out[f"{tag}_rollout_param_{name}"] = p.detach().numpy()
            out[f"{tag}_rollout_grad_{name}"] = p.grad.numpy()
        # ---- the optimisation --------------------------------------------------------------------------------------------------------
        args = lambda p: dict(batched_observed_pos_m=torch.from_numpy(p.copy()), batched_valid_mask=torch.from_numpy(valid.copy()),  # noqa: E731
                              batched_observed_yaw_angle_rad=torch.from_numpy(yaw.copy()), batched_vehicle_length_m=torch.from_numpy(length),
                              time_between_frames_s=0.1)
        for iters in (1, 3, 30):
            res = ts.smooth_track_bike_model(**args(pos), max_iters=iters, return_losses=True)
            out[f"{tag}_{iters}_pos"], out[f"{tag}_{iters}_rot"], out[f"{tag}_{iters}_velo"] = (r.detach().numpy().copy() for r in res[:3])
            out[f"{tag}_{iters}_first_loss"] = res[3][0]["per_batch_loss"]
            out[f"{tag}_{iters}_last_loss"] = res[3][-1]["per_batch_loss"]
            out[f"{tag}_{iters}_evaluations"] = np.array(len(res[3]))
            print(tag, iters, "evaluations", len(res[3]), "loss", res[3][0]["per_batch_loss"].mean(), "->", res[3][-1]["per_batch_loss"].mean())
        p2 = pos.copy()
        p2[0, 5, 0] += 1e-6
        res2 = ts.smooth_track_bike_model(**args(p2), max_iters=30)
        out[f"{tag}_sensitivity"] = np.array(float((res2[0].detach() - torch.from_numpy(out[f"{tag}_30_pos"])).abs().max()))
        print(tag, "sensitivity", out[f"{tag}_sensitivity"])
    np.savez_compressed(os.path.join(HERE, "bike_model_reference.npz"), **out)


if __name__ == "__main__":
    sys.path.insert(0, "/root/reference")
    main()
