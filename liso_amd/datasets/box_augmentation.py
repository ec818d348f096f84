"""Box-snippet augmentation with the sweep, the snippet database and every per-point step on the device.

Mirror of `LidarDataset.create_augmented_sample_from_box_snippet_db`
(liso/datasets/torch_dataset_commons.py:1531-1776; caller :1805-1830): same arguments, same result dictionary (`gt.boxes`,
`pcl_ta.{pcl, pillar_coors}`, `pcl_full_w_ground_ta`, `pcl_full_no_ground_ta`, `<flow_source>.flow_ta_tb`,
`<train_on_box_source>.{boxes, prediscovered_boxes, centermaps_*}`), tensors on the sweep's device.

The reference runs this per sample in DataLoader workers: a scikit-image disk dilation of the 512^2 occupancy map (~317 probes
per cell), a Python loop over the pasted objects, numpy re-pillarisation and numpy target rendering.  With the detector step at
a few ms that is the input bottleneck.  Here:
  * the free-location mask, the selection of the drawn cells, the snippet gather + pose + flow + box speed are the kernels of
    include/liso_augment.h (liso_amd/csrc/box_augment.hip);
  * re-pillarisation is `voxelize_sample` on the device, target maps are `render_center_targets` (include/liso_detector.h);
  * the random draws are made on the host from numpy's / torch's global generators IN THE REFERENCE'S ORDER (a few dozen
    scalars and one index permutation per object), so a seeded run reproduces the reference's sample
    (tests/golden/box_augment_reference.npz); they need ONE device read per sample, the number of free cells.

Kept reference behaviour (see oracle/box_augment.py for the list): no z shift of the pasted points (`t_z=None`), float32 sin /
cos of the heading (evaluated on the host like the reference), float64 pose arithmetic rounded once to float32.
"""
import ctypes

import numpy as np
import torch

from liso_amd import _lib as L
from liso_amd.datasets.targets import render_center_targets
from liso_amd.datasets.torch_dataset_commons import voxelize_sample
from liso_amd.kabsch.shape_utils import Shape
from liso_amd.utils.bev_utils import get_bev_setup_params


class BoxSnippetDb:
    """The snippet database of `load_sanitize_box_augmentation_database` (liso/tracker/augm_box_db_utils.py:13-59) resident in
    HBM: all snippets concatenated ([T,4] float32, box coordinates + intensity) with host-side offsets (the draws that need the
    snippet sizes are made on the host), the boxes as a host `Shape`, and the per-point LiDAR rows for the ray-drop variant."""

    def __init__(self, db, device):
        pcls = [np.asarray(p, np.float32) for p in db["pcl_in_box_cosy"]]
        assert len(pcls) > 0 and all(p.ndim == 2 and p.shape[1] == 4 for p in pcls)
        self.counts = np.array([p.shape[0] for p in pcls], np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.counts)]).astype(np.int64)
        self.device = torch.device(device)
        self.points = torch.from_numpy(np.concatenate(pcls, 0)).to(self.device).contiguous()
        boxes = db["boxes"]
        if not isinstance(boxes, Shape):  # the saved form: the Shape's attribute dictionary of numpy arrays (:41)
            boxes = Shape(**boxes)
        self.boxes = boxes.clone().cpu() if torch.is_tensor(boxes.pos) else boxes.to_tensor()  # (Shape.cpu / .to work in place)
        assert self.boxes.pos.shape[0] == len(pcls), (self.boxes.pos.shape, len(pcls))
        self.lidar_rows = [np.asarray(r) for r in db["lidar_rows"]] if "lidar_rows" in db else None
        self.box_T_sensor = db.get("box_T_sensor")

    def __len__(self):
        return len(self.counts)


@torch.no_grad()
def free_location_mask(pillar_coors, grid_hw, radius):
    """-> (free uint8 [H,W] (1 = an object centre may go here), row_free_prefix int32 [H+1]); reference :1538-1557"""
    L.require_cuda(pillar_coors)
    H, W = int(grid_hw[0]), int(grid_hw[1])
    coors = pillar_coors.to(torch.int32).contiguous()
    assert coors.dim() == 2 and coors.shape[1] == 2, coors.shape
    dev = coors.device
    free = torch.empty((H, W), dtype=torch.uint8, device=dev)
    prefix = torch.empty(H + 1, dtype=torch.int32, device=dev)
    ws_bytes = int(L.lib().liso_bev_free_mask_workspace_bytes(H, W))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("bev_free_mask", lambda: L.lib().liso_bev_free_mask(
            L.ptr(coors), coors.shape[0], H, W, int(radius), L.ptr(free), L.ptr(prefix), L.ptr(ws), ws_bytes, L.stream_ptr())),
            "bev_free_mask")
    return free, prefix


@torch.no_grad()
def select_free_cells(free, prefix, compact_idx):
    """row-major index among the free cells -> flat cell index (int32 [K]); reference :1558-1563"""
    H, W = free.shape
    idx = torch.as_tensor(compact_idx, dtype=torch.int64).to(free.device).contiguous()
    out = torch.empty(idx.shape[0], dtype=torch.int32, device=free.device)
    with torch.cuda.device(free.device):
        L.check(L.lib().liso_bev_select_free_cells(L.ptr(free), L.ptr(prefix), H, W, L.ptr(idx), idx.shape[0], L.ptr(out),
                                                   L.stream_ptr()), "bev_select_free_cells")
    return out


@torch.no_grad()
def paste_snippets(db, src_index, out_offsets, pose, flow_rand, vmin, vmax, want_flow=True):
    """-> (points float32 [n,4], flow float32 [n,3] or None, box speed float32 [K]); reference :1597-1690"""
    dev = db.points.device
    n, k = int(out_offsets[-1]), len(out_offsets) - 1
    src = torch.as_tensor(src_index, dtype=torch.int64).to(dev).contiguous()
    offs = torch.as_tensor(out_offsets, dtype=torch.int64).to(dev).contiguous()
    pose_d = torch.as_tensor(pose, dtype=torch.float64).reshape(k, 12).to(dev).contiguous()
    rnd = torch.as_tensor(flow_rand, dtype=torch.float64).reshape(n, 3).to(dev).contiguous()
    pts = torch.empty((n, 4), dtype=torch.float32, device=dev)
    flow = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_flow else None
    velo = torch.empty(k, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("snippet_paste", lambda: L.lib().liso_snippet_paste(
            L.ptr(db.points), db.points.shape[0], L.ptr(src), L.ptr(offs), L.ptr(pose_d), L.ptr(rnd), float(vmin), float(vmax), k,
            L.ptr(pts), L.ptr(flow) if flow is not None else None, L.ptr(velo), L.stream_ptr())), "snippet_paste")
    return pts, flow, velo


class BoxAugmenter:
    """Holds what the reference keeps on the dataset object: the BEV set-up (`get_bev_setup_params`), the augmentation block of
    the config (`cfg.data.augmentation.boxes`, liso_config.yml:56-66) and the snippet database."""

    def __init__(self, cfg, db: BoxSnippetDb, need_flow=True, centermaps_output_grid_size=None, reference_draws=True):
        """`reference_draws=False` replaces the ONE draw whose cost grows with the grid -- the reference's
        `np.random.choice(arange(num_free), k, replace=False)`, a permutation of all ~200k free cells (~2 ms of the ~3 ms a
        sample takes) -- by k distinct `randint` draws: same distribution, different generator stream (no fixture parity)."""
        self.reference_draws = reference_draws
        self.cfg = cfg
        self.box_augm_cfg = cfg.data.augmentation.boxes
        self.box_augm_db = db
        self.need_flow = need_flow
        (self.bev_range_m_np, self.img_grid_size_np, self.bev_pixel_per_meter_res_np, self.pcl_bev_center_coords_homog_np,
         _) = get_bev_setup_params(cfg)
        self.centermaps_output_grid_size = (np.asarray(centermaps_output_grid_size) if centermaps_output_grid_size is not None
                                            else self.img_grid_size_np // 4)
        self._centers_xy = np.ascontiguousarray(self.pcl_bev_center_coords_homog_np[..., :2]).reshape(-1, 2)
        if getattr(cfg.data, "limit_pillar_height", False):
            self.height_range_m = tuple(float(v) for v in cfg.data.pillar_height_range_m)
        else:
            self.height_range_m = (-np.inf, np.inf)

    # -- reference :1778-1806: the two ray-drop variants only select rows (and consume the generator) ------------------------------
    @staticmethod
    def layer_based_raydrop_augm(per_pt_row_idxs):
        keep_every_nth_row = np.random.choice([1, 2, 3])
        keep_row_start_idx = np.random.choice(np.arange(0, keep_every_nth_row))
        return ((per_pt_row_idxs - keep_row_start_idx) % keep_every_nth_row) == 0

    @staticmethod
    def _distinct_randint(n, k):
        picked = []
        while len(picked) < k:
            v = int(np.random.randint(0, n))
            if v not in picked:
                picked.append(v)
        return np.array(picked, np.int64)

    def _draw_point_selection(self, obj_idx):
        """rows of snippet `obj_idx` that are pasted, in output order (reference :1598-1650)"""
        db, bc = self.box_augm_db, self.box_augm_cfg
        n = int(db.counts[obj_idx])
        if bc.use_raydrop_augm:
            if db.lidar_rows is None:
                raise ValueError("use_raydrop_augm needs the `lidar_rows` of the snippet database")
            keep = self.layer_based_raydrop_augm(db.lidar_rows[obj_idx].astype(np.int32))
            if np.count_nonzero(keep) == 0:
                return np.arange(n)
            # (the reference then evaluates its resolution ray-drop on the kept points and discards the result, :1612-1632: two draws)
            np.random.choice([600, 900, 1200, 1500])
            np.random.choice([1, 2])
            return np.nonzero(keep)[0]
        if bc.max_points_dropout != 0.0:
            num_keep = max(1, int(n * (1.0 - np.random.rand() * bc.max_points_dropout)))
            return np.random.choice(np.arange(start=0, stop=n, step=1, dtype=int), num_keep, replace=False)
        return np.arange(n)

    @torch.no_grad()
    def create_augmented_sample_from_box_snippet_db(self, src_trgt_time_delta_s, sample_data_ta, prediscovered_boxes: Shape = None):
        bc, db = self.box_augm_cfg, self.box_augm_db
        dev = db.device
        num_augm_objs = np.random.randint(low=1, high=bc.max_num_objs + 1)
        size_single_pillar_m = 1 / self.bev_pixel_per_meter_res_np
        min_dist = bc.setdefault("min_obj_center_dist_from_occupied_pillars_m", 2.0) if hasattr(bc, "setdefault") else 2.0
        assert size_single_pillar_m.shape == (2,)
        radius = max(3, int(min_dist / size_single_pillar_m.mean()))
        pillar_coors = sample_data_ta["pcl_ta"]["pillar_coors"].to(dev)
        free, prefix = free_location_mask(pillar_coors, self.img_grid_size_np, radius)
        num_free = int(prefix[-1].item())  # the one device read: the draw below is over arange(num_free)
        if num_free < num_augm_objs:
            raise ValueError(f"{num_augm_objs} objects to place but only {num_free} free BEV cells")  # (np.random.choice raises too)
        if self.reference_draws:
            augm_loc_idxs = np.random.choice(np.arange(num_free), size=num_augm_objs, replace=False)
        else:
            augm_loc_idxs = self._distinct_randint(num_free, num_augm_objs)
        flat_cells = select_free_cells(free, prefix, augm_loc_idxs).cpu().numpy()
        augm_box_locations_xy = torch.from_numpy(self._centers_xy[flat_cells])
        augm_box_locations_xy += (0.5 - torch.rand_like(augm_box_locations_xy)) * torch.from_numpy(size_single_pillar_m)
        obj_idxs = np.random.choice(np.arange(len(db)), size=num_augm_objs, replace=True)
        box_dims = db.boxes[obj_idxs].dims
        box_z_pos_old = db.boxes[obj_idxs].pos[..., [2]]
        box_z_pos_new = 0.5 * (torch.rand((num_augm_objs, 1)) - 0.5) + box_z_pos_old
        box_rot = 2 * np.pi * (torch.rand((num_augm_objs, 1)) - 0.5)
        box_pos = torch.cat([augm_box_locations_xy, box_z_pos_new], dim=-1)
        # sensor_T_box of torch_compose_matrix(t_x, t_y, theta_z, t_z=None): rotation about z, translation (x, y, 0)
        sin, cos = torch.sin(box_rot[:, 0]).double().numpy(), torch.cos(box_rot[:, 0]).double().numpy()
        tx, ty = box_pos[:, 0].double().numpy(), box_pos[:, 1].double().numpy()
        extra_boxes = Shape(pos=box_pos, dims=box_dims, rot=box_rot, probs=torch.ones_like(box_rot))

        pose = np.zeros((num_augm_objs, 3, 4), np.float64)
        sel, rands, out_offsets = [], [], [0]
        for i, obj_idx in enumerate(obj_idxs):
            rows = self._draw_point_selection(obj_idx)
            flip_x = 1 if np.random.rand() < 0.5 else -1
            flip_y = 1 if np.random.rand() < 0.5 else -1
            scale_x = 1.0 - bc.max_scale_delta * (2 * np.random.rand() - 1.0)
            scale_y = 1.0 - bc.max_scale_delta * (2 * np.random.rand() - 1.0)
            scale_z = 1.0 - bc.max_scale_delta * (2 * np.random.rand() - 1.0)
            fx, fy = flip_x * scale_x, flip_y * scale_y
            pose[i] = [[cos[i] * fx, -sin[i] * fy, 0.0, tx[i]], [sin[i] * fx, cos[i] * fy, 0.0, ty[i]], [0.0, 0.0, scale_z, 0.0]]
            sel.append(db.offsets[obj_idx] + rows)
            out_offsets.append(out_offsets[-1] + rows.shape[0])
            # (the flow draws of this object follow its flips / scales in the generator's stream, :1672-1677)
            rands.append(np.random.rand(rows.shape[0], 3))
        src_index = np.concatenate(sel)
        flow_rand = np.concatenate(rands, 0)
        extra_pcl, extra_flows, velo = paste_snippets(db, src_index, np.array(out_offsets, np.int64), pose, flow_rand,
                                                      bc.min_artificial_obj_velo, bc.max_artificial_obj_velo, want_flow=self.need_flow)
        extra_boxes.velo = velo[:, None].cpu()

        if prediscovered_boxes is not None:
            extra_boxes = extra_boxes.cat(prediscovered_boxes.clone().cpu(), dim=0)
            assert torch.all(extra_boxes.probs == 1.0)
        else:
            prediscovered_boxes = Shape.createEmpty().to_tensor()

        cat = lambda t: torch.cat([t.to(dev), extra_pcl], dim=0)  # noqa: E731
        pcl_ta = cat(sample_data_ta["pcl_ta"]["pcl"])
        augm = {
            "gt": {"boxes": extra_boxes},
            "pcl_full_w_ground_ta": cat(sample_data_ta["pcl_full_w_ground_ta"]),
            "pcl_full_no_ground_ta": cat(sample_data_ta["pcl_full_no_ground_ta"]),
            "src_trgt_time_delta_s": torch.tensor(src_trgt_time_delta_s),
        }
        if "odom_ta_tb" in sample_data_ta["gt"]:
            augm["gt"]["odom_ta_tb"] = sample_data_ta["gt"]["odom_ta_tb"].clone()
        flow_source = self.cfg.data.flow_source
        if flow_source not in augm:
            augm[flow_source] = {}
        # pillarize_bev (:1147-1163): cells of all points, points outside the BEV range (or the pillar height limits) dropped from
        # the `pcl_ta` / `flow_ta_tb` / `pillar_coors_ta` entries
        coors, in_range = voxelize_sample(pcl_ta, self.bev_range_m_np, self.img_grid_size_np, self.height_range_m)
        augm["pcl_ta"] = {"pcl": pcl_ta[in_range], "pillar_coors": coors[in_range]}
        if self.need_flow:
            flow_all = torch.cat([sample_data_ta[flow_source]["flow_ta_tb"].to(dev), extra_flows], dim=0)
            augm[flow_source]["flow_ta_tb"] = flow_all[in_range]

        if self.cfg.network.name not in ("pointrcnn", "pointpillars"):
            if self.cfg.loss.supervised.centermaps.confidence_target != "gaussian":
                raise NotImplementedError(self.cfg.loss.supervised.centermaps.confidence_target)
            b = extra_boxes.clone().to(dev)
            maps = render_center_targets(b.pos[None].float(), b.dims[None].float(), b.rot[None].float(), b.valid[None],
                                         self.centermaps_output_grid_size, self.bev_range_m_np)
            centermaps = {f"centermaps_{k}": v[0] for k, v in maps.items()}
        else:
            centermaps = {}
        augm[self.cfg.data.train_on_box_source] = {"boxes": extra_boxes, "prediscovered_boxes": prediscovered_boxes, **centermaps}
        return augm

    def create_augmented_sample_from_flow_cluster_detector_and_box_snippet_db(self, src_trgt_time_delta_s, sample_data_ta):
        """reference :1805-1830: the boxes already mined for this sample (`sample[train_on_box_source]["boxes"]`) ride along with the
        pasted ones when cluster supervision or box augmentation is active; without a snippet database only the time delta is returned"""
        sup = self.cfg.loss.supervised
        if ("supervised_on_clusters" in sup and sup.supervised_on_clusters.active) or self.cfg.data.augmentation.boxes.active:
            prediscovered_boxes = sample_data_ta.get(self.cfg.data.train_on_box_source, {}).get("boxes", None)
        else:
            assert "mined" not in sample_data_ta, sample_data_ta["mined"].keys()
            prediscovered_boxes = None
        if self.box_augm_db is None:
            return {"src_trgt_time_delta_s": torch.tensor(src_trgt_time_delta_s)}
        return self.create_augmented_sample_from_box_snippet_db(src_trgt_time_delta_s, sample_data_ta, prediscovered_boxes=prediscovered_boxes)
