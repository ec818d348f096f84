"""Generates tests/golden/box_refinement_reference.npz from the reference's own python:
  liso.box_fitting.box_fitting.fit_2d_box_modest(..., fit_method="closeness_to_edge")        (box_fitting.py:93-141,242-258)
  liso.tracker.tracking.perform_local_box_refinement + set_box_size_keep_closest_point_constant (tracking.py:239-260,2004-2133)
liso.tracker.tracking as a whole cannot be imported here (see make_tracking_golden.py): the two functions are compiled at generation
time from the reference file's own text with the reference's Shape / homogenize_pcl / fit_2d_box_modest as their globals; nothing of
the reference is stored.  Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_box_refinement_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import _Anything, cfg, import_with_stubs  # noqa: E402
from make_tracking_golden import function_from_reference_file  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def car_points(g, n, length, width, yaw, cx, cy):
    """points on two visible sides of a car-sized rectangle (an L shape) plus a few interior returns"""
    t = g.uniform(-0.5, 0.5, n)
    side = g.uniform(size=n) < 0.6
    x = np.where(side, t * length, 0.5 * length) + g.normal(0, 0.03, n)
    y = np.where(side, -0.5 * width, t * width) + g.normal(0, 0.03, n)
    c, s = np.cos(yaw), np.sin(yaw)
    return np.stack([cx + c * x - s * y, cy + s * x + c * y, g.uniform(-1.5, 0.0, n)], -1)


def main():
    def _imp():
        from liso.box_fitting.box_fitting import fit_2d_box_modest
        from liso.kabsch.shape_utils import Shape
        from liso.utils.torch_transformation import homogenize_pcl
        return fit_2d_box_modest, Shape, homogenize_pcl

    fit_2d_box_modest, Shape, homogenize_pcl = import_with_stubs(_imp)

    class FlowClusterDetector:  # (isinstance target of the quantile choice, tracking.py:2016-2019)
        pass

    class BoxLearner:
        pass

    env = {"Shape": Shape, "homogenize_pcl": homogenize_pcl, "fit_2d_box_modest": fit_2d_box_modest, "torch": torch, "np": np,
           "FlowClusterDetector": FlowClusterDetector, "BoxLearner": BoxLearner}
    path = "/root/reference/liso/tracker/tracking.py"
    env["set_box_size_keep_closest_point_constant"] = function_from_reference_file(path, "set_box_size_keep_closest_point_constant", env)
    refine = function_from_reference_file(path, "perform_local_box_refinement", env)

    g = np.random.default_rng(21)
    out = {}
    # --- the rectangle fit on its own -----------------------------------------------------------------------------------------
    for i, (n, L, W, yaw) in enumerate([(300, 4.5, 1.9, 0.3), (40, 4.0, 1.8, 1.2), (3, 2.0, 1.0, -0.4), (1, 1.0, 1.0, 0.0), (800, 1.0, 3.0, 2.5)]):
        pts = car_points(g, n, L, W, yaw, g.uniform(-20, 20), g.uniform(-20, 20))
        center, length, width, ry = fit_2d_box_modest(pts, fit_method="closeness_to_edge")
        out[f"fit{i}_points"] = pts
        out[f"fit{i}_result"] = np.array([center[0], center[1], length, width, ry], np.float64)
    # --- whole tracks ------------------------------------------------------------------------------------------------------------
    for tag, (age, start, fit_rot, fit_pos, flow_cluster, n_bg) in {"t0": (6, 2, True, True, True, 4000), "t1": (4, 0, True, False, False, 1500),
                                                                    "t2": (5, 1, False, False, True, 800)}.items():
        c = cfg({"data": {"tracking_cfg": {"fit_box_to_points": {"fit_rot": fit_rot, "fit_pos": fit_pos, "fitting_dims_bloat_factor": 1.2}}}})
        clouds, pos, dims, rot = [], [], [], []
        x0, y0, yaw0 = g.uniform(-15, 15), g.uniform(-15, 15), g.uniform(-3, 3)
        for t in range(start + age):
            cx, cy, yaw = x0 + 0.8 * t * np.cos(yaw0), y0 + 0.8 * t * np.sin(yaw0), yaw0 + 0.02 * t
            car = car_points(g, int(g.integers(0, 250)) if t != start + 1 else 0, 4.4, 1.9, yaw, cx, cy)  # one frame without returns
            bg = np.concatenate([g.uniform(-40, 40, (n_bg, 2)), g.uniform(-2, 1, (n_bg, 1))], -1)
            cloud = np.concatenate([car, bg], 0).astype(np.float32)
            clouds.append(torch.from_numpy(np.concatenate([cloud, g.uniform(0, 1, (cloud.shape[0], 1)).astype(np.float32)], -1)))
            if t >= start:
                pos.append([cx + g.normal(0, 0.3), cy + g.normal(0, 0.3), -0.8])
                dims.append([4.4 + g.normal(0, 0.4), 1.9 + g.normal(0, 0.2), 1.6 + g.normal(0, 0.1)])
                rot.append([yaw + g.normal(0, 0.15)])
        boxes = Shape(pos=torch.tensor(pos, dtype=torch.float32), dims=torch.tensor(dims, dtype=torch.float32),
                      rot=torch.tensor(rot, dtype=torch.float32), probs=torch.ones(age, 1))
        out[f"{tag}_meta"] = np.array([age, start, fit_rot, fit_pos, flow_cluster], np.float64)
        for k in ("pos", "dims", "rot"):
            out[f"{tag}_in_{k}"] = getattr(boxes, k).numpy().copy()
        out[f"{tag}_cloud_sizes"] = np.array([cl.shape[0] for cl in clouds])
        out[f"{tag}_clouds"] = torch.cat(clouds, 0).numpy()
        res = refine(c, FlowClusterDetector() if flow_cluster else BoxLearner(), clouds, boxes.clone(), age, start)
        for k in ("pos", "dims", "rot"):
            out[f"{tag}_out_{k}"] = getattr(res, k).numpy()
        print(tag, np.abs(out[f"{tag}_out_rot"] - out[f"{tag}_in_rot"]).max(), np.abs(out[f"{tag}_out_pos"] - out[f"{tag}_in_pos"]).max())
    np.savez_compressed(os.path.join(HERE, "box_refinement_reference.npz"), **out)
    print("arrays:", len(out))


if __name__ == "__main__":
    main()
