"""`Shape`: the box container of the hot path.  Mirror of liso/kabsch/shape_utils.py (API subset used by the
rows of SURVEY.md section 8; shapely contours and plotting corners are out of scope).

Fields (reference :19-93): pos[...,{2,3}], dims[...,{1,2,3}], rot[...,1], probs[...,1], velo[...,{1,2,3}],
valid[...] bool, class_id[...,1] int32, difficulty[...,1] int32.
"""
import pprint

import numpy as np
import torch

from liso_amd.utils.torch_transformation import (
    homogenize_pcl,
    numpy_compose_matrix,
    torch_compose_matrix,
    torch_decompose_matrix,
)

UNKNOWN_CLASS_ID = torch.iinfo(torch.int32).max  # reference :15
INVALID_CLASS_ID = UNKNOWN_CLASS_ID - 1  # reference :16


def _is_t(x):
    return torch.is_tensor(x)


class Shape:
    _numeric_float_keys = ("pos", "dims", "rot", "probs", "velo")
    _numeric_int_keys = ("class_id", "difficulty")
    _keys = _numeric_float_keys + _numeric_int_keys + ("valid",)

    def __init__(self, pos, dims, rot, probs, velo=None, valid=None, class_id=None, difficulty=None):
        assert pos.shape[-1] in (1, 2, 3), pos.shape
        assert dims.shape[-1] in (1, 2, 3), dims.shape
        assert probs.shape[-1] == 1, probs.shape
        t = _is_t(probs)
        if not t and not isinstance(probs, np.ndarray):
            raise NotImplementedError(type(probs))
        self.pos, self.dims, self.rot, self.probs = pos, dims, rot, probs
        if valid is None:  # reference :41-50
            valid = torch.ones_like(probs[..., 0], dtype=torch.bool) if t else np.ones_like(probs[..., 0], dtype=bool)
        self.valid = valid
        if velo is None:  # reference :52-60
            velo = torch.zeros_like(probs) if t else np.zeros_like(probs)
        assert velo.shape[-1] in (1, 2, 3), velo.shape
        self.velo = velo
        if class_id is None:  # reference :62-76
            class_id = (UNKNOWN_CLASS_ID * torch.ones_like(pos[..., :1], dtype=torch.int32) if t
                        else UNKNOWN_CLASS_ID * np.ones_like(pos[..., :1], dtype=np.int32))
        assert class_id.shape[-1] == 1, class_id.shape
        self.class_id = class_id
        if difficulty is None:  # reference :78-88 (ones for tensors, zeros for arrays)
            difficulty = (torch.ones_like(pos[..., :1], dtype=torch.int) if t
                          else np.zeros_like(pos[..., :1], dtype=np.int32))
        assert difficulty.shape[-1] == 1, difficulty.shape
        self.difficulty = difficulty
        assert len(self.valid.shape) == len(self.probs.shape) - 1
        assert len(self.valid.shape) == len(self.dims.shape) - 1
        if self.rot is not None:
            assert len(self.valid.shape) == len(self.rot.shape) - 1

    # ---- construction -------------------------------------------------------------------------------------
    @staticmethod
    def createEmpty():  # reference :95-106
        return Shape(pos=np.empty((0, 3)), dims=np.empty((0, 3)), rot=np.empty((0, 1)), probs=np.empty((0, 1)),
                     valid=np.zeros((0), dtype=bool), class_id=np.zeros((0, 1), dtype=np.int32),
                     difficulty=np.zeros((0, 1), dtype=np.int32), velo=np.empty((0, 1)))

    @property
    def shape(self):
        return self.valid.shape

    @staticmethod
    def from_list_of_shapes(shapes_list, numeric_padding_value=np.nan, int_padding_value=INVALID_CLASS_ID):
        """reference :112-140 -- pad a list of [K_i,...] shapes into one [B,Kmax,...] batch."""
        pad = torch.nn.utils.rnn.pad_sequence
        if all(s.valid.shape == () for s in shapes_list):
            out = {"valid": torch.stack([s.valid for s in shapes_list])}
        else:
            out = {"valid": pad([s.valid for s in shapes_list], batch_first=True, padding_value=False)}
        for k in Shape._numeric_float_keys:
            out[k] = pad([getattr(s, k) for s in shapes_list], batch_first=True, padding_value=numeric_padding_value)
        for k in Shape._numeric_int_keys:
            out[k] = pad([getattr(s, k) for s in shapes_list], batch_first=True, padding_value=int_padding_value)
        return Shape(**out)

    def to_tensor(self):
        return Shape(**{k: torch.from_numpy(v) for k, v in self.__dict__.items()})

    # ---- conversions --------------------------------------------------------------------------------------
    def _map(self, fn, skip=()):
        return Shape(**{k: (fn(v) if (v is not None and k not in skip) else v) for k, v in self.__dict__.items()})

    def detach(self):
        return self._map(lambda v: v.detach() if _is_t(v) else v)

    def clone(self):
        return self._map(lambda v: v.clone() if _is_t(v) else v.copy())

    def to(self, device_or_dtype):
        """reference :181-197 -- in place; dtype casts leave valid/class_id/difficulty alone."""
        if isinstance(device_or_dtype, (torch.device, str)):
            dont = ()
        elif isinstance(device_or_dtype, torch.dtype) or device_or_dtype in (np.float32, np.float64):
            dont = ("valid",) + self._numeric_int_keys
        else:
            raise NotImplementedError(f"Don't know what to do with {device_or_dtype}")
        for k, v in self.__dict__.items():
            if k in dont or v is None:
                continue
            self.__dict__[k] = v.to(device_or_dtype) if _is_t(v) else v.astype(device_or_dtype)
        return self

    def cpu(self):
        for k, v in self.__dict__.items():
            self.__dict__[k] = v.to("cpu")
        return self

    def numpy(self):
        return self.clone()._map(lambda v: v.detach().cpu().numpy() if _is_t(v) else v)

    # ---- indexing -----------------------------------------------------------------------------------------
    def __getitem__(self, key):
        return Shape(**{k: (v[key] if v is not None else None) for k, v in self.__dict__.items()})

    def drop_padding_boxes(self):
        """reference :243-269"""
        if len(self.shape) == 1:
            m = self.valid.clone() if _is_t(self.valid) else self.valid.copy()
            return Shape(**{k: ((v.clone() if _is_t(v) else np.copy(v))[m] if v is not None else None)
                            for k, v in self.__dict__.items()})
        return Shape.from_list_of_shapes([self[i].drop_padding_boxes() for i in range(self.shape[0])])

    def into_list_of_shapes(self):
        return [self[i].drop_padding_boxes() for i in range(self.shape[0])]

    def cat(self, other, dim):
        return Shape(**{k: (torch.cat([v, getattr(other, k)], dim=dim) if v is not None else None)
                        for k, v in self.__dict__.items()})

    def change_order_confidence_descending(self):
        assert len(self.pos.shape) == 2, "can't handle batched inputs"
        order = torch.squeeze(torch.argsort(self.probs, dim=0, descending=True), dim=-1)
        for k, v in self.__dict__.items():
            if v is not None:
                self.__dict__[k] = v[order]

    def set_padding_val_to(self, value=0.0, int_value=INVALID_CLASS_ID):
        """reference :439-462"""
        for keys, val in ((Shape._numeric_float_keys, value), (Shape._numeric_int_keys, int_value)):
            for k in keys:
                v = self.__dict__[k]
                if _is_t(v):
                    self.__dict__[k] = torch.where(self.valid[..., None], v, val)  # (scalar overload: no host->device copy)
                else:
                    self.__dict__[k] = np.where(self.valid[..., None], v, val)

    def assert_attr_shapes_compatible(self):
        for k, v in self.__dict__.items():
            if v is not None and k != "valid":
                assert v.shape[:-1] == self.valid.shape, (k, v.shape, self.valid.shape)

    def __str__(self):
        return pprint.pformat(self.__dict__)

    # ---- geometry -----------------------------------------------------------------------------------------
    def get_poses(self, check_finite=True):
        """reference :271-319 -- sensor_T_box as fp64 [.., 4, 4] (yaw about z only).  `check_finite=False` (extension) skips the
        reference's `assert all(isfinite(rot))`, which is a device->host read."""
        unb = len(self.pos.shape) == 2
        pos = self.pos[None, ...] if unb else self.pos
        rot = None if self.rot is None else (self.rot[None, ...] if unb else self.rot)
        assert len(pos.shape) == 3, self.pos.shape
        if _is_t(self.pos):
            tz = None if pos.shape[-1] == 2 else pos[..., 2].to(torch.double)
            if rot is None or rot.shape[-1] == 0:
                th = torch.zeros_like(pos[..., 0], dtype=torch.double)
            else:
                if check_finite:
                    assert torch.all(torch.isfinite(rot))
                th = rot[..., 0].to(torch.double)
            pose = torch_compose_matrix(pos[..., 0].to(torch.double), pos[..., 1].to(torch.double), th, t_z=tz)
        else:
            tz = None if pos.shape[-1] == 2 else pos[..., 2].astype(np.float64)
            if rot is None or rot.shape[-1] == 0:
                th = np.zeros_like(pos[..., 0], dtype=np.float64)
            else:
                assert np.all(np.isfinite(rot))
                th = rot[..., 0].astype(np.float64)
            pose = numpy_compose_matrix(pos[..., 0].astype(np.float64), pos[..., 1].astype(np.float64), th, t_z=tz)
        return pose[0] if unb else pose

    @staticmethod
    def get_bottom_corner_idxs():
        return (0, 1, 4, 5)  # reference :373-375

    def get_box_corners(self):
        """reference :377-436 (tensor branch): the 8 corners in sensor coordinates, fp64 [.., 8, 3], and the edge list; corner order
        front-right-bottom, front-left-bottom, front-left-top, front-right-top, then the same for the rear"""
        assert self.dims.shape[-1] == 3 and self.pos.shape[-1] == 3, (self.dims.shape, self.pos.shape)
        edges = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))
        signs = [(1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)]
        unit = 0.5 * torch.tensor(signs, dtype=self.dims.dtype, device=self.dims.device)
        corners = unit * self.dims[..., None, :]
        homog = torch.cat([corners, torch.ones_like(corners[..., :1])], dim=-1)
        pose = self.get_poses()
        return torch.einsum("...ij,...cj->...ci", pose, homog.to(pose.dtype))[..., :3], edges

    def transform(self, new_T_old):
        """reference :472-486"""
        out = self.clone()
        pos_new, rot_new = torch_decompose_matrix(new_T_old @ self.get_poses())
        out.pos, out.rot = pos_new, rot_new
        out.assert_attr_shapes_compatible()
        return out

    @torch.no_grad()
    def get_points_in_box_bool_mask(self, pcl, box_dims_bloat_factor=1.0, return_points_in_box_coords=False):
        """reference :488-538 -- [.., N, K] inside test in box coordinates.  Device tensors go through
        liso_points_in_boxes_f32 (include/liso_tracking.h), which never materialises [N,K,4]; host arrays (the reference's
        data-loader use) and `return_points_in_box_coords` keep the reference's formulation."""
        assert pcl.shape[-1] == 3, pcl.shape
        assert len(self.shape) == len(pcl.shape) - 1, (self.shape, pcl.shape)
        if _is_t(pcl) and pcl.is_cuda and not return_points_in_box_coords and self.dims.shape[-1] == 3 and self.pos.shape[-1] == 3:
            from liso_amd.tracker.box_points import FP32_PRODUCT, dense_boxes, points_in_boxes
            unb = len(self.shape) == 1
            boxes7 = dense_boxes(self)
            res = points_in_boxes(boxes7[None] if unb else boxes7, pcl[None] if unb else pcl, want_mask=True, want_count=False,
                                  precision=FP32_PRODUCT, dims_bloat=box_dims_bloat_factor)
            return res["mask"][0] if unb else res["mask"]
        sensor_T_box = self.get_poses()
        homog = homogenize_pcl(pcl[..., :3])
        dims = box_dims_bloat_factor * self.dims
        if _is_t(homog):
            pts = torch.einsum("...kij,...nj->...nki", torch.linalg.inv(sensor_T_box).to(torch.float), homog)
            inside = torch.all(torch.abs(pts[..., 0:3]) < 0.5 * dims, dim=-1)
        else:
            pts = np.einsum("...kij,...nj->...nki", np.linalg.inv(sensor_T_box), homog)
            inside = np.all(np.abs(pts[..., 0:3]) < 0.5 * dims, axis=-1)
        return (inside, pts) if return_points_in_box_coords else inside


def is_boxes_clearly_in_bev_range(boxes, bev_range_m):
    """reference :549-560"""
    assert len(bev_range_m) == 2, bev_range_m
    xp = torch if _is_t(boxes.pos) else np
    box_xy = xp.abs(boxes.pos[..., :2]) - boxes.dims[..., [0]] / 2
    ok = xp.abs(box_xy) < bev_range_m / 2
    return torch.all(ok, dim=-1) if xp is torch else np.all(ok, axis=-1)


def extract_box_motion_transform_without_sensor_odometry(pred_boxes_a, fg_kabsch_trafos, bg_kabsch_trafo, check=True):
    """reference :583-605 -- b0_dT_b1 = inv(T_box) inv(T_bg) T_fg T_box (all fp64).  `check=False` (extension): the same LU
    inverses without torch.linalg.inv's singularity check (a device->host read per call) and without get_poses' finiteness assert."""
    s0_T_box0 = pred_boxes_a.get_poses(check_finite=check)
    inv = torch.linalg.inv if check else (lambda m: torch.linalg.inv_ex(m).inverse)
    return inv(s0_T_box0) @ inv(bg_kabsch_trafo) @ (fg_kabsch_trafos @ s0_T_box0)


def extract_motion_in_pred_box_coordinates(pred_boxes_a, fg_kabsch_trafos, bg_kabsch_trafo, check=True):
    """reference :563-580"""
    return torch_decompose_matrix(
        extract_box_motion_transform_without_sensor_odometry(pred_boxes_a, fg_kabsch_trafos, bg_kabsch_trafo, check=check))
