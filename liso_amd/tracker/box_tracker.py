"""Per-sequence box stores of the mining loop.  Mirror of liso/tracker/box_tracker.py:8-126 (`NotATracker`) and of the parts
liso/tracker/global_box_tracker.py shares with it (:38-76 `update`, :469-514 the getters): same method names, arguments and returned
values.  A sequence's boxes live on the host here, as in the reference (a few dozen boxes per frame, bookkeeping with data-dependent
shapes); the per-frame work that produces them -- detector / flow clustering, NMS, flow propagation -- runs on the device upstream."""
from typing import Dict, List

import torch

from liso_amd.kabsch.shape_utils import Shape
from liso_amd.tracker.tracking_helpers import aggregate_odometry_to_world_poses


class SequenceBoxStore:
    """what both trackers share: the per-frame detections in sensor coordinates, the odometry chain, world / sensor views of the
    result and the track-id queries"""

    def __init__(self) -> None:
        self.boxes_sensor_ti = []
        self.sti_T_stii = []
        self.per_box_extra_attributes_dict = []
        self.w_Ts_sti = None
        self.max_track_id_counter = 0
        self.has_tracked = False

    def _store(self, boxes_tii_s: Shape, odom_stii_stiii, per_box_extra_attributes_tii):
        assert len(boxes_tii_s.pos.shape) == 2, ("batching not supported", boxes_tii_s.pos.shape)
        assert len(odom_stii_stiii.shape) == 2, ("batching not supported", odom_stii_stiii.shape)
        assert torch.all(boxes_tii_s.valid), "can't handle invalid boxes -> drop_invalid_boxes()"
        self.boxes_sensor_ti.append(boxes_tii_s.detach().cpu())
        self.sti_T_stii.append(odom_stii_stiii.detach().cpu())
        self.per_box_extra_attributes_dict.append(per_box_extra_attributes_tii)

    def _to_world(self):
        """every frame's detections in world coordinates (frame t under the product of the first t odometries)"""
        self.w_Ts_sti = aggregate_odometry_to_world_poses(self.sti_T_stii)
        return [boxes.transform(self.w_Ts_sti[t]) for t, boxes in enumerate(self.boxes_sensor_ti)]

    def get_boxes_in_world_coordinates(self):
        return self.boxes_world_ti

    def get_boxes_in_sensor_coordinates_at_each_timestamp(self):
        assert self.has_tracked, "need to run tracking first"
        self.boxes_sensor_ti = [bw.clone().transform(torch.linalg.inv(w_T_s)) for bw, w_T_s in zip(self.boxes_world_ti, self.w_Ts_sti)]
        return self.boxes_sensor_ti

    def get_extra_attributes_at_each_timestamp(self):
        return self.per_box_extra_attributes_dict

    def get_all_unique_track_ids_and_lengths(self):
        return torch.unique(torch.concat(self.track_ids, dim=0), return_counts=True)

    def get_min_max_track_id(self):
        ids, _ = self.get_all_unique_track_ids_and_lengths()
        if ids.size()[0] > 0:
            return ids.min(), ids.max()
        return torch.tensor(0).to(ids.device), torch.tensor(0).to(ids.device)

    def get_ids_lengths_of_longest_tracks(self):
        ids, lens = self.get_all_unique_track_ids_and_lengths()
        order = torch.argsort(lens, descending=True)
        return ids[order], lens[order]

    def get_box_indices_start_time_for_track_id(self, track_id):
        padded = torch.nn.utils.rnn.pad_sequence(self.track_ids, batch_first=True, padding_value=-1)
        timestamps, box_idxs = torch.where(padded == track_id)
        return box_idxs, timestamps[0]


class NotATracker(SequenceBoxStore):
    """every detection is its own track (reference :8-79)"""

    def __init__(self) -> None:
        super().__init__()
        self.detection_ids_ti = []
        self.max_det_id_counter = 0

    def update(self, boxes_tii_s: Shape, predicted_box_poses_stiii: None, predicted_box_poses_sti: None, odom_stii_stiii: torch.Tensor,
               per_box_extra_attributes_tii: List[Dict[str, str]] = None):
        self._store(boxes_tii_s, odom_stii_stiii, per_box_extra_attributes_tii)
        n = boxes_tii_s.valid.shape[0]
        det_ids = self.max_det_id_counter + 1 + torch.arange(start=0, end=n, device=boxes_tii_s.valid.device, dtype=torch.long)
        if n > 0:
            self.max_det_id_counter = torch.max(det_ids)
        self.detection_ids_ti.append(det_ids)

    def run_tracker(self):
        self.max_track_id_counter = 0
        self.boxes_world_ti = self._to_world()
        self.track_ids = self.detection_ids_ti
        self.has_tracked = True
