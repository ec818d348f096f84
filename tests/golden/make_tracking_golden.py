"""Generates tests/golden/tracking_reference.npz from the reference's own python:
  liso.datasets.torch_dataset_commons.get_points_in_boxes_mask            (:1902-1935, torch branch, fp64 transform)
  liso.kabsch.shape_utils.Shape.get_points_in_box_bool_mask                (:488-538, torch branch, fp32 transform)
  liso.tracker.tracking.propagate_boxes_forward_using_flow                 (:2168-2211)
  liso.kabsch.box_groundtruth_matching_iou.match_boxes_by_descending_confidence_iou (:8-68, greedy)
The modules' unrelated third-party imports that are absent from this image are stubbed with empty modules (names only).
liso.tracker.tracking as a whole cannot be imported here (its import tree reaches mmdet3d / mmcv class hierarchies), so the
ONE function needed from it is compiled at generation time from the reference file's own text (ast: the FunctionDef node of
propagate_boxes_forward_using_flow) with the reference's own Shape / extract_box_motion_transform_without_sensor_odometry
as its globals -- the reference's code is executed where it lies, nothing of it is stored in this repository.
The matching function obtains its IoU matrix from the reference's CUDA extension, which cannot run here: the matrix is an
INPUT of the fixture (random values incl. ties and NaN) and is handed to the function in place of box_iou_matrix.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_tracking_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402  (also puts /root/reference on sys.path)

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")  # logging only; absent from this image


def function_from_reference_file(path, name, env):
    import ast
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[ast.ImportFrom(module="typing", names=[ast.alias(name="*")], level=0), node], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), env)
    return env[name]


def scene(g, K, N, R=40.0):
    pos = np.concatenate([g.uniform(-0.4 * R, 0.4 * R, (K, 2)), g.uniform(-1.2, -0.6, (K, 1))], -1)
    dims = np.stack([g.uniform(3.0, 5.5, K), g.uniform(1.5, 2.4, K), g.uniform(1.4, 2.0, K)], -1)
    rot = g.uniform(-np.pi, np.pi, (K, 1))
    # half of the points are drawn around the boxes (so they are hit), the rest uniformly
    near = pos[g.integers(0, K, N // 2)] + g.normal(0.0, 1.2, (N // 2, 3))
    far = np.concatenate([g.uniform(-0.5 * R, 0.5 * R, (N - N // 2, 2)), g.uniform(-2.0, 1.0, (N - N // 2, 1))], -1)
    pts = np.concatenate([near, far], 0)
    g.shuffle(pts, axis=0)
    return pos.astype(np.float32), dims.astype(np.float32), rot.astype(np.float32), pts.astype(np.float32)


def main():
    def _imp():
        from liso.datasets.torch_dataset_commons import get_points_in_boxes_mask
        from liso.kabsch.shape_utils import Shape
        from liso.kabsch.shape_utils import extract_box_motion_transform_without_sensor_odometry
        import liso.kabsch.box_groundtruth_matching_iou as matching
        return get_points_in_boxes_mask, Shape, extract_box_motion_transform_without_sensor_odometry, matching

    get_points_in_boxes_mask, Shape, extract_motion, matching = import_with_stubs(_imp)
    propagate = function_from_reference_file("/root/reference/liso/tracker/tracking.py", "propagate_boxes_forward_using_flow",
                                             {"torch": torch, "Shape": Shape,
                                              "extract_box_motion_transform_without_sensor_odometry": extract_motion})
    g = np.random.default_rng(7)
    out = {}
    for tag, (K, N) in {"a": (12, 6000), "b": (37, 20000), "c": (1, 500)}.items():
        pos, dims, rot, pts = scene(g, K, N)
        boxes = Shape(pos=torch.from_numpy(pos), dims=torch.from_numpy(dims), rot=torch.from_numpy(rot),
                      probs=torch.ones(K, 1), valid=torch.ones(K, dtype=torch.bool))
        homog = torch.cat([torch.from_numpy(pts), torch.ones(N, 1)], -1)
        m64 = get_points_in_boxes_mask(boxes, homog)
        m32 = boxes.get_points_in_box_bool_mask(torch.from_numpy(pts))
        m32_bloat = boxes.get_points_in_box_bool_mask(torch.from_numpy(pts), box_dims_bloat_factor=1.25)
        flow = (g.normal(0.0, 0.5, (N, 3)) + np.array([1.0, -0.5, 0.0])).astype(np.float32)
        valid = g.uniform(size=N) > 0.1
        th = 0.02
        odom = np.eye(4)
        odom[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        odom[:3, 3] = [0.9, 0.05, 0.01]
        fg, _, bg, warped, st1 = propagate(boxes[None], torch.from_numpy(pts)[None], torch.from_numpy(valid)[None],
                                           torch.from_numpy(flow)[None], torch.from_numpy(odom), "cpu")
        out.update({f"{tag}_pos": pos, f"{tag}_dims": dims, f"{tag}_rot": rot, f"{tag}_pts": pts, f"{tag}_flow": flow,
                    f"{tag}_valid": valid, f"{tag}_odom": odom,
                    f"{tag}_mask64": np.packbits(m64.numpy(), axis=0), f"{tag}_mask32": np.packbits(m32.numpy(), axis=0),
                    f"{tag}_mask32_bloat": np.packbits(m32_bloat.numpy(), axis=0),
                    f"{tag}_fg": fg.numpy(), f"{tag}_bg": bg.numpy(), f"{tag}_warped": warped.numpy(), f"{tag}_st1": st1.numpy()})

    # greedy matching: the IoU matrix is fixture input
    for tag, (n_gt, n_pred) in {"m0": (9, 14), "m1": (40, 25), "m2": (1, 6), "m3": (5, 1), "m4": (70, 130)}.items():
        iou = g.uniform(0.0, 1.0, (n_gt, n_pred)).astype(np.float32)
        iou[g.uniform(size=iou.shape) < 0.5] = 0.0               # most pairs do not overlap
        if n_gt > 4 and n_pred > 4:
            iou[2, :] = iou[1, :]                                 # exact ties between two ground-truth rows
            iou[3, 2] = np.nan
        conf = g.uniform(0.05, 1.0, (n_pred, 1)).astype(np.float32)
        matching.box_iou_matrix = lambda a, b, mode, _m=iou: torch.from_numpy(_m)
        dummy = lambda n: Shape(pos=torch.zeros(n, 3), dims=torch.ones(n, 3), rot=torch.zeros(n, 1),  # noqa: E731
                                probs=torch.ones(n, 1), valid=torch.ones(n, dtype=torch.bool))
        pred = dummy(n_pred)
        pred.probs = torch.from_numpy(conf)
        for thr in (0.3, 0.5):
            ig, ip, d, pm, gm = matching.match_boxes_by_descending_confidence_iou(dummy(n_gt), pred, thr)
            out.update({f"{tag}_{thr}_idx_gt": ig, f"{tag}_{thr}_idx_pred": ip, f"{tag}_{thr}_dists": np.asarray(d, dtype=np.float32),
                        f"{tag}_{thr}_pred_mask": pm, f"{tag}_{thr}_gt_mask": gm})
        out.update({f"{tag}_iou": iou, f"{tag}_conf": conf})
    np.savez_compressed(os.path.join(HERE, "tracking_reference.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items() if k.startswith(("a_", "m0_"))})
    print(os.path.getsize(os.path.join(HERE, "tracking_reference.npz")) / 1e3, "kB")


if __name__ == "__main__":
    main()
