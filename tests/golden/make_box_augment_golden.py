"""Generates tests/golden/box_augment_reference.npz from the reference's own python:
  liso.datasets.torch_dataset_commons.LidarDataset.create_augmented_sample_from_box_snippet_db   (:1531-1776)
called UNBOUND on a plain attribute holder that carries exactly the dataset attributes the method reads (the BEV set-up of
liso.utils.bev_utils.get_bev_setup_params, the box-augmentation config block of liso_config.yml:56-66, a small snippet
database in the layout of liso.tracker.augm_box_db_utils.load_sanitize_box_augmentation_database) and the reference's own
`pillarize_bev` / `voxelize_sample` / `move_pcl_pillar_coors_to_subdict` / `select_centermaps_target_confidence` bound to it.
`skimage.morphology.disk` / `binary_dilation` are evaluated by the real scikit-image of /opt/conda/bin/python3.9 (the main
interpreter of this image has none) through a subprocess; other absent third-party modules are stubbed with empty modules.

Every case stores the inputs, the numpy / torch seeds, and the method's outputs.  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_box_augment_golden.py
"""
import os
import pickle
import subprocess
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from make_targets_golden import cfg, import_with_stubs  # noqa: E402

SKIMAGE_PY = "/opt/conda/bin/python3.9"
_SK_SNIPPET = (
    "import pickle,sys\n"
    "from skimage.morphology import disk, binary_dilation\n"
    "fn, args = pickle.load(sys.stdin.buffer)\n"
    "out = disk(*args) if fn == 'disk' else binary_dilation(args[0], args[1])\n"
    "pickle.dump(out, sys.stdout.buffer, protocol=2)\n"
)


def _skimage(fn, *args):
    r = subprocess.run([SKIMAGE_PY, "-c", _SK_SNIPPET], input=pickle.dumps((fn, args), protocol=2), stdout=subprocess.PIPE, check=True)
    return pickle.loads(r.stdout)


def disk(radius):
    return _skimage("disk", int(radius))


def binary_dilation(image, footprint=None):
    return _skimage("binary_dilation", np.asarray(image), np.asarray(footprint))


def make_db(g, Shape, n_obj):
    pcls, rows, T = [], [], []
    dims = np.stack([g.uniform(3.0, 5.0, n_obj), g.uniform(1.5, 2.2, n_obj), g.uniform(1.4, 1.8, n_obj)], -1).astype(np.float32)
    for i in range(n_obj):
        n = int(g.integers(11, 90))
        p = (g.uniform(-0.5, 0.5, (n, 3)) * dims[i]).astype(np.float32)
        pcls.append(np.concatenate([p, g.uniform(0, 1, (n, 1)).astype(np.float32)], -1))
        rows.append(g.integers(0, 64, n).astype(np.uint8))
        th = g.uniform(-np.pi, np.pi)
        M = np.eye(4)
        M[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        M[:3, 3] = [g.uniform(-20, 20), g.uniform(-20, 20), g.uniform(-1.5, -0.5)]
        T.append(np.linalg.inv(M))
    boxes = Shape(pos=torch.from_numpy(np.stack([np.linalg.inv(t)[:3, 3] for t in T]).astype(np.float32)), dims=torch.from_numpy(dims),
                  rot=torch.from_numpy(g.uniform(-np.pi, np.pi, (n_obj, 1)).astype(np.float32)), probs=torch.ones(n_obj, 1))
    return {"pcl_in_box_cosy": pcls, "lidar_rows": rows, "boxes": boxes, "box_T_sensor": torch.from_numpy(np.stack(T))}


def main():
    def _imp():
        import liso.datasets.torch_dataset_commons as tdc
        from liso.kabsch.shape_utils import Shape
        from liso.utils.bev_utils import get_bev_setup_params
        return tdc, Shape, get_bev_setup_params

    tdc, Shape, get_bev_setup_params = import_with_stubs(_imp)
    tdc.disk, tdc.binary_dilation = disk, binary_dilation
    LD = tdc.LidarDataset
    out = {}
    cases = {
        # tag: (grid, range, n scene points, db objects, max_num_objs, max_points_dropout, need_flow, prediscovered, seed)
        "a": (128, 60.0, 1500, 6, 5, 0.25, True, 0, 11),
        "b": (256, 100.0, 6000, 12, 15, 0.25, False, 3, 12),
        "c": (64, 40.0, 300, 3, 2, 0.0, True, 0, 13),
    }
    for tag, (G, R, n_pts, n_db, max_objs, dropout, need_flow, n_pre, seed) in cases.items():
        g = np.random.default_rng(seed)
        c = cfg({
            "data": {"bev_range_m": [R, R], "img_grid_size": [G, G], "flow_source": "gt" if tag == "c" else "slim_flow",
                     "train_on_box_source": "mined", "limit_pillar_height": False,
                     "augmentation": {"boxes": {"active": True, "max_num_objs": max_objs, "min_artificial_obj_velo": 1.0,
                                                "max_artificial_obj_velo": 3.0, "max_scale_delta": 0.2,
                                                "max_points_dropout": dropout, "use_raydrop_augm": False}}},
            "network": {"name": "centerpoint"},
            "loss": {"supervised": {"centermaps": {"confidence_target": "gaussian"}}},
            "box_prediction": {"dimensions_representation": {"method": "predict_abs_size"}, "rotation_representation": {"method": "vector"},
                               "position_representation": {"method": "local_relative_offset"}, "activations": {"dims": "softplus"}},
        })
        self = types.SimpleNamespace()
        (self.bev_range_m_np, self.img_grid_size_np, self.bev_pixel_per_meter_res_np, self.pcl_bev_center_coords_homog_np,
         _) = get_bev_setup_params(c)
        self.cfg = c
        self.box_augm_cfg = c.data.augmentation.boxes
        self.height_range_m_np = np.array([-np.inf, np.inf], np.float32)
        self.centermaps_output_grid_size = self.img_grid_size_np // 4
        self.need_flow = need_flow
        self.box_augm_db = make_db(g, Shape, n_db)
        for name in ("pillarize_bev", "voxelize_sample", "move_pcl_pillar_coors_to_subdict", "select_centermaps_target_confidence",
                     "layer_based_raydrop_augm", "resolution_raydrop_augmentation"):
            setattr(self, name, types.MethodType(getattr(LD, name), self))
        self.get_sample_data_downsample_keys = LD.get_sample_data_downsample_keys

        # the scene: points of a few walls and blobs, all inside the BEV range; some pasted objects land outside the z limits never
        pcl = np.concatenate([g.uniform(-0.5 * R, 0.5 * R, (n_pts, 2)) * g.choice([0.2, 0.6, 0.98], (n_pts, 1)),
                              g.uniform(-2.0, 1.0, (n_pts, 1)), g.uniform(0, 1, (n_pts, 1))], -1).astype(np.float32)
        pcl_t = torch.from_numpy(pcl)
        coors, in_range = self.voxelize_sample(pcl_t)
        assert bool(in_range.all())
        sample = {
            "pcl_ta": {"pcl": pcl_t, "pillar_coors": coors},
            "pcl_full_w_ground_ta": torch.from_numpy(np.concatenate([pcl, pcl[:50]], 0)),
            "pcl_full_no_ground_ta": pcl_t.clone(),
            "gt": {"odom_ta_tb": torch.eye(4, dtype=torch.float64)},
            c.data.flow_source: {"flow_ta_tb": torch.from_numpy(g.normal(size=(n_pts, 3)).astype(np.float32))},
        }
        pre = None
        if n_pre:
            pre = Shape(pos=torch.from_numpy(g.uniform(-10, 10, (n_pre, 3)).astype(np.float32)),
                        dims=torch.from_numpy(g.uniform(1.5, 4.5, (n_pre, 3)).astype(np.float32)),
                        rot=torch.from_numpy(g.uniform(-3, 3, (n_pre, 1)).astype(np.float32)), probs=torch.ones(n_pre, 1))
            pre.velo = torch.from_numpy(g.uniform(0, 3, (n_pre, 1)).astype(np.float32))
        np.random.seed(seed)
        torch.manual_seed(seed)
        res = LD.create_augmented_sample_from_box_snippet_db(self, 0.1, sample, prediscovered_boxes=pre)

        out[f"{tag}_meta"] = np.array([G, R, max_objs, dropout, float(need_flow), seed, n_pre], np.float64)
        out[f"{tag}_flow_source"] = np.array(c.data.flow_source)
        out[f"{tag}_in_pcl"] = pcl
        out[f"{tag}_in_coors"] = coors.numpy()
        out[f"{tag}_in_flow"] = sample[c.data.flow_source]["flow_ta_tb"].numpy()
        db = self.box_augm_db
        out[f"{tag}_db_points"] = np.concatenate(db["pcl_in_box_cosy"], 0)
        out[f"{tag}_db_counts"] = np.array([p.shape[0] for p in db["pcl_in_box_cosy"]], np.int64)
        out[f"{tag}_db_rows"] = np.concatenate(db["lidar_rows"], 0)
        out[f"{tag}_db_box_T_sensor"] = db["box_T_sensor"].numpy()
        for k in ("pos", "dims", "rot", "probs"):
            out[f"{tag}_db_box_{k}"] = getattr(db["boxes"], k).numpy()
        if pre is not None:
            for k in ("pos", "dims", "rot", "probs", "velo"):
                out[f"{tag}_pre_{k}"] = getattr(pre, k).numpy()
        # outputs
        out[f"{tag}_out_pcl"] = res["pcl_ta"]["pcl"].numpy()
        out[f"{tag}_out_coors"] = res["pcl_ta"]["pillar_coors"].numpy()
        out[f"{tag}_out_pcl_full_w_ground"] = res["pcl_full_w_ground_ta"].numpy()
        out[f"{tag}_out_pcl_full_no_ground"] = res["pcl_full_no_ground_ta"].numpy()
        if need_flow:
            out[f"{tag}_out_flow"] = res[c.data.flow_source]["flow_ta_tb"].numpy()
        b = res["gt"]["boxes"]
        for k in ("pos", "dims", "rot", "probs", "velo", "valid"):
            out[f"{tag}_out_box_{k}"] = np.asarray(getattr(b, k))
        mined = res[c.data.train_on_box_source]
        for k, v in mined.items():
            if k.startswith("centermaps_"):
                out[f"{tag}_out_{k}"] = v.numpy()
        out[f"{tag}_out_n_prediscovered"] = np.array(mined["prediscovered_boxes"].pos.shape[0])
        # the free-location mask on its own (the first thing the method computes)
        occ = np.zeros((G, G), bool)
        occ[coors[:, 0].numpy(), coors[:, 1].numpy()] = True
        radius = max(3, int(2.0 / (1.0 / self.bev_pixel_per_meter_res_np).mean()))
        out[f"{tag}_free_mask"] = ~binary_dilation(occ, footprint=disk(radius))
        out[f"{tag}_radius"] = np.array(radius)
        print(tag, {k: v.shape for k, v in out.items() if k.startswith(f"{tag}_out")})
    np.savez_compressed(os.path.join(HERE, "box_augment_reference.npz"), **out)


if __name__ == "__main__":
    main()
