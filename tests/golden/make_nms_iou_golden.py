"""Generates tests/golden/nms_iou_reference.npz from the reference's own python boundary:
  liso.utils.nms_iou.{convert_shapes_to_dense_3d, rotate_nms_pcdet, iou_based_nms, box_iou_matrix (iou_bev / iou_3d),
                      boxes_iou_bev, perform_nms_on_shapes, hard_limit_detections}          (nms_iou.py:10-282)
  iou3d_nms.iou3d_nms_utils.{to_pcdet, boxes_iou_bev, boxes_iou3d_gpu, nms_gpu, nms_normal_gpu}  (iou3d_nms_utils.py:12-106)
and tests/golden/voxelize_pcl_reference.npz (row B4) from
  liso.datasets.nuscenes.analyse_boxes.voxelize_pcl                                          (analyse_boxes.py:6-26).
The native module `iou3d_nms_cuda` those files import cannot run here (CUDA).  It is replaced, for the generation only, by
a 5-function module with the same names and call conventions whose numbers come from oracle/_ref -- the reference's
UNMODIFIED iou3d_cpu.cpp compiled where it lies (oracle/Makefile) -- plus the greedy sweep of iou3d_nms.cpp:113-132 over
`IoU > thresh` bits (oracle/iou3d_oracle.c, itself pinned bitwise to oracle/_ref).  `torch.cuda.FloatTensor` /
`Tensor.cuda()` are mapped to their CPU counterparts while the reference functions run.  Nothing of the reference is stored.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_nms_iou_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_targets_golden import import_with_stubs  # noqa: E402  (also puts /root/reference on sys.path)

from oracle import iou3d as O  # noqa: E402


def install_native_stub():
    m = types.ModuleType("iou3d_nms_cuda")

    def boxes_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(O.ref_boxes_overlap_bev(a.numpy(), b.numpy())))
        return 1

    def boxes_iou_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(O.ref_boxes_iou_bev(a.numpy(), b.numpy())))
        return 1

    def nms_gpu(boxes, keep, thresh):
        k = O.nms_from_iou(O.ref_boxes_iou_bev(boxes.numpy(), boxes.numpy()), thresh)
        keep[:len(k)] = torch.from_numpy(k)
        return len(k)

    def nms_normal_gpu(boxes, keep, thresh):
        k = O.nms_normal(boxes.numpy(), thresh)
        keep[:len(k)] = torch.from_numpy(k)
        return len(k)

    m.boxes_overlap_bev_gpu, m.boxes_iou_bev_gpu, m.nms_gpu, m.nms_normal_gpu = (boxes_overlap_bev_gpu, boxes_iou_bev_gpu,
                                                                                  nms_gpu, nms_normal_gpu)
    m.boxes_iou_bev_cpu = boxes_iou_bev_gpu
    sys.modules["iou3d_nms_cuda"] = m


class _cpu_cuda:
    """while active: torch.cuda.FloatTensor(size) -> CPU float tensor, Tensor.cuda() -> self"""

    def __enter__(self):
        self.saved = (torch.cuda.FloatTensor, torch.Tensor.cuda)
        torch.cuda.FloatTensor = lambda size: torch.empty(tuple(size), dtype=torch.float32)
        torch.Tensor.cuda = lambda self, *a, **k: self

    def __exit__(self, *exc):
        torch.cuda.FloatTensor, torch.Tensor.cuda = self.saved


def random_shape(Shape, g, n, with_z=True, pad=0):
    pos = np.concatenate([g.uniform(-30, 30, (n, 2)), g.uniform(-1.6, -0.4, (n, 1))], -1).astype(np.float32)
    dims = np.stack([g.uniform(2, 6, n), g.uniform(1, 2.5, n), g.uniform(1.2, 2.2, n)], -1).astype(np.float32)
    rot = g.uniform(-np.pi, np.pi, (n, 1)).astype(np.float32)
    probs = g.uniform(0, 1, (n, 1)).astype(np.float32)
    valid = np.ones(n, bool)
    if pad:
        valid[g.choice(n, pad, replace=False)] = False
    return dict(pos=pos if with_z else pos[:, :2], dims=dims if with_z else dims[:, :2], rot=rot, probs=probs, valid=valid)


def main():
    install_native_stub()

    def _imp():
        import liso.utils.nms_iou as nms_iou
        from liso.kabsch.shape_utils import Shape
        return nms_iou, Shape

    nms_iou, Shape = import_with_stubs(_imp)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_iou3d_nms_utils", "/root/reference/iou3d_nms/iou3d_nms_utils.py")
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)

    def S(d):
        return Shape(**{k: torch.from_numpy(v) for k, v in d.items()})

    g = np.random.default_rng(4)
    out = {}
    with _cpu_cuda(), torch.no_grad():
        # --- box_iou_matrix, both modes, 3-D and 2-D (pos/dims without z) shapes, incl. an empty side ---------------------
        for tag, (na, nb, with_z) in {"m3": (60, 90, True), "m2": (40, 25, False), "me": (0, 12, True)}.items():
            a, b = random_shape(Shape, g, na, with_z), random_shape(Shape, g, nb, with_z)
            if na:
                b["pos"][:min(na, nb) // 2] = a["pos"][:min(na, nb) // 2] + g.normal(0, 0.5, a["pos"][:min(na, nb) // 2].shape).astype(np.float32)
            for k, v in a.items():
                out[f"{tag}_a_{k}"] = v
            for k, v in b.items():
                out[f"{tag}_b_{k}"] = v
            out[f"{tag}_dense_a"] = nms_iou.convert_shapes_to_dense_3d(S(a).clone()).numpy()
            out[f"{tag}_iou_bev"] = nms_iou.box_iou_matrix(S(a), S(b), iou_mode="iou_bev").numpy()
            if with_z:
                out[f"{tag}_iou_3d"] = nms_iou.box_iou_matrix(S(a), S(b), iou_mode="iou_3d").numpy()
        # --- iou_based_nms / rotate_nms_pcdet with pre / post top-k -----------------------------------------------------------
        for tag, (n, pre, post, thr) in {"n1": (300, None, None, 0.1), "n2": (300, 120, 25, 0.1), "n3": (200, 1000, 100, 0.3),
                                         "n4": (1, None, None, 0.1)}.items():
            s = random_shape(Shape, g, n)
            s["pos"][:, :2] *= 0.5  # denser: more suppression
            for k, v in s.items():
                out[f"{tag}_{k}"] = v
            out[f"{tag}_cfg"] = np.array([-1 if pre is None else pre, -1 if post is None else post, thr], np.float64)
            out[f"{tag}_keep"] = nms_iou.iou_based_nms(S(s), thr, pre_nms_max_boxes=pre, post_nms_max_boxes=post).numpy()
        # --- perform_nms_on_shapes on a padded batch --------------------------------------------------------------------------
        B, K = 3, 160
        batch = [random_shape(Shape, g, K, pad=p) for p in (0, 40, 159)]
        for d in batch:
            d["pos"][:, :2] *= 0.4
        stacked = {k: np.stack([d[k] for d in batch]) for k in batch[0]}
        for k, v in stacked.items():
            out[f"p_{k}"] = v
        res = nms_iou.perform_nms_on_shapes(S(stacked), max_num_boxes=30, overlap_threshold=0.1, pre_nms_max_num_boxes=100)
        for k in ("pos", "dims", "rot", "probs", "valid"):
            out[f"p_out_{k}"] = getattr(res, k).numpy()
        # --- iou3d_nms_utils (OpenPCDet wrappers) -----------------------------------------------------------------------------
        a, b = random_shape(Shape, g, 70), random_shape(Shape, g, 50)
        b["pos"][:30] = a["pos"][:30] + g.normal(0, 0.6, (30, 3)).astype(np.float32)
        da = torch.from_numpy(np.concatenate([a["pos"], a["dims"], a["rot"]], -1))
        db = torch.from_numpy(np.concatenate([b["pos"], b["dims"], b["rot"]], -1))
        out["u_a"], out["u_b"], out["u_scores"] = da.numpy(), db.numpy(), a["probs"][:, 0]
        out["u_to_pcdet"] = utils.to_pcdet(da.clone()).numpy()
        out["u_iou_bev"] = utils.boxes_iou_bev(da, db).numpy()
        out["u_iou3d"] = utils.boxes_iou3d_gpu(da.clone(), db.clone()).numpy()
        out["u_nms"] = utils.nms_gpu(da, torch.from_numpy(a["probs"][:, 0]), 0.1, pre_maxsize=60)[0].numpy()
        out["u_nms_normal"] = utils.nms_normal_gpu(da, torch.from_numpy(a["probs"][:, 0]), 0.1)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "nms_iou_reference.npz"), **out)
    print("nms_iou fixture:", {k: v.shape for k, v in out.items() if "keep" in k or "iou" in k or k.startswith("u_nms")})

    # ---- B4: dataset-side pillar coordinates (integer: bit-exact) + collate padding -----------------------------------------
    def _imp2():
        from liso.datasets.nuscenes.analyse_boxes import voxelize_pcl
        from liso.datasets.torch_dataset_commons import LidarDataset, collate_list_data
        return voxelize_pcl, LidarDataset, collate_list_data

    voxelize_pcl, LidarDataset, collate_list_data = import_with_stubs(_imp2)
    vx = {}
    for tag, (n, R, G, hr) in {"a": (5000, 100.0, 512, (-2.0, 1.0)), "b": (3000, 51.2, 640, (-np.inf, np.inf)),
                               "c": (2000, 120.0, 920, (-2.0, 1.0))}.items():
        pcl = np.concatenate([g.uniform(-0.55 * R, 0.55 * R, (n, 2)), g.uniform(-3.0, 2.0, (n, 1)), g.uniform(0, 1, (n, 1))], -1)
        pcl = pcl.astype(np.float32)
        # boundary cases: exactly on the range limits, on pillar edges, one ulp inside / outside, tiny negatives (which the
        # int32 truncation maps to pillar 0: analyse_boxes.py:11-17)
        f = np.float32
        edge = np.array([-R / 2, R / 2, np.nextafter(f(R / 2), f(0)), np.nextafter(f(-R / 2), f(0)), np.nextafter(f(-R / 2), f(-R)),
                         0.0, -0.0, f(R / G), f(-R / G), np.nextafter(f(0), f(-1)), f(R / 2 - R / G), f(-R / 2 - 0.5 * R / G),
                         f(-R / 2 - 0.999 * R / G), f(-R / 2 - 1.001 * R / G)], np.float32)
        k = len(edge)
        pcl[:k, 0], pcl[k:2 * k, 1] = edge, edge
        pcl[2 * k:3 * k, 0], pcl[2 * k:3 * k, 1] = edge, edge[::-1]
        pcl[3 * k:3 * k + 4, 2] = np.array([-2.0, 1.0, np.nextafter(f(-2.0), f(0)), np.nextafter(f(1.0), f(0))])
        ds = types.SimpleNamespace(bev_range_m_np=np.array([R, R], np.float32), img_grid_size_np=np.array([G, G]).astype(np.int32),
                                   height_range_m_np=np.array(hr, np.float32))
        coors, in_range = LidarDataset.voxelize_sample(ds, pcl.copy())  # torch_dataset_commons.py:975-987
        vx[f"{tag}_pcl"], vx[f"{tag}_cfg"] = pcl, np.array([R, G, hr[0], hr[1]], np.float64)
        vx[f"{tag}_pillar_coors"], vx[f"{tag}_in_range"] = np.asarray(coors), np.asarray(in_range)
        # the function itself on a torch tensor and on a numpy array with the caller's dtypes (analyse_boxes.py:6-26)
        rng3, grid3 = np.append(ds.bev_range_m_np, np.array(1000.0)), np.append(ds.img_grid_size_np, np.array(1))
        c_t, m_t = voxelize_pcl(torch.from_numpy(pcl.copy()), torch.from_numpy(rng3), torch.from_numpy(grid3))
        vx[f"{tag}_torch_coors"], vx[f"{tag}_torch_in"] = c_t.numpy(), m_t.numpy()
    # collate: two samples of different length -> NaN / -1 padding + validity mask (torch_dataset_commons.py:370-431)
    def sample(n, seed):
        gg = np.random.default_rng(seed)
        pc = gg.uniform(-20, 20, (n, 4)).astype(np.float32)
        return {"pcl_ta": {"pcl": torch.from_numpy(pc), "pillar_coors": torch.from_numpy(gg.integers(0, 512, (n, 2)).astype(np.int32))},
                "gt": {"flow_ta_tb": torch.from_numpy(gg.normal(0, 1, (n, 3)).astype(np.float32)),
                       "odom_ta_tb": torch.from_numpy(np.eye(4) + gg.normal(0, 0.01, (4, 4)))},
                "pcl_full_no_ground_ta": torch.from_numpy(pc[: n // 2]),
                "src_trgt_time_delta_s": torch.tensor(0.1)}
    samples = [sample(700, 1), sample(1000, 2), sample(450, 3)]
    for i, smp in enumerate(samples):
        vx[f"col_in{i}_pcl"], vx[f"col_in{i}_coors"] = smp["pcl_ta"]["pcl"].numpy(), smp["pcl_ta"]["pillar_coors"].numpy()
        vx[f"col_in{i}_flow"], vx[f"col_in{i}_odom"] = smp["gt"]["flow_ta_tb"].numpy(), smp["gt"]["odom_ta_tb"].numpy()
    col = collate_list_data(samples)
    vx["col_pcl"], vx["col_valid"], vx["col_coors"] = (col["pcl_ta"]["pcl"].numpy(), col["pcl_ta"]["pcl_is_valid"].numpy(),
                                                        col["pcl_ta"]["pillar_coors"].numpy())
    vx["col_flow"], vx["col_odom"], vx["col_dt"] = col["gt"]["flow_ta_tb"].numpy(), col["gt"]["odom_ta_tb"].numpy(), col["src_trgt_time_delta_s"].numpy()
    vx["col_full_lens"] = np.array([t.shape[0] for t in col["pcl_full_no_ground_ta"]])
    np.savez_compressed(os.path.join(HERE, "voxelize_pcl_reference.npz"), **vx)
    print("voxelize_pcl fixture:", {k: (v.shape, v.dtype) for k, v in vx.items() if k.startswith("a_") or k.startswith("col_")})


if __name__ == "__main__":
    main()
