"""TEST INFRASTRUCTURE ONLY: numpy restatement of the mining / tracking / validation inner loops.
Pinned by tests/golden/tracking_reference.npz, generated from the reference's own python
(tests/golden/make_tracking_golden.py); checked in tests/test_oracle_tracking.py."""
import numpy as np


def box_poses(pos, rot):
    """Shape.get_poses (liso/kabsch/shape_utils.py:271-319): sensor_T_box fp64 [K,4,4], yaw about z"""
    K = pos.shape[0]
    T = np.tile(np.eye(4), (K, 1, 1))
    th = rot[:, 0].astype(np.float64)
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = np.cos(th), -np.sin(th), np.sin(th), np.cos(th)
    T[:, :3, 3] = pos.astype(np.float64)
    return T


def points_in_boxes_mask(pos, dims, rot, pts):
    """get_points_in_boxes_mask, torch branch (liso/datasets/torch_dataset_commons.py:1902-1935): fp64 transform, fp32
    comparison -> bool [N,K]"""
    homog = np.concatenate([pts[:, :3].astype(np.float64), np.ones((pts.shape[0], 1))], -1)
    pcl_box = np.einsum("kij,nj->nki", np.linalg.inv(box_poses(pos, rot)), homog).astype(np.float32)
    return np.all(np.abs(pcl_box[:, :, 0:3]) < np.float32(0.5) * dims[None].astype(np.float32), axis=-1)


def points_in_box_bool_mask(pos, dims, rot, pts, box_dims_bloat_factor=1.0):
    """Shape.get_points_in_box_bool_mask, torch branch (liso/kabsch/shape_utils.py:488-523): inverse rounded to fp32,
    fp32 product -> bool [N,K]"""
    homog = np.concatenate([pts[:, :3].astype(np.float32), np.ones((pts.shape[0], 1), np.float32)], -1)
    inv32 = np.linalg.inv(box_poses(pos, rot)).astype(np.float32)
    pcl_box = np.einsum("kij,nj->nki", inv32, homog)
    rel = (np.float32(box_dims_bloat_factor) * dims.astype(np.float32))
    return np.all(np.abs(pcl_box[:, :, 0:3]) < np.float32(0.5) * rel[None], axis=-1)


def mean_flow_per_box(pos, dims, rot, pts, valid, flow):
    """liso/tracker/tracking.py:2176-2185 -> fp32 [K,3]; the denominator counts every in-box point, valid or not"""
    m = points_in_box_bool_mask(pos, dims, rot, pts).astype(np.float32)  # [N,K]
    num = (flow[:, None, :].astype(np.float32) * valid[:, None, None].astype(np.float32) * m[:, :, None]).sum(axis=0)
    return num / np.clip(m.sum(axis=0), 1.0, None)[:, None]


def propagate_boxes_forward_using_flow(pos, dims, rot, pts, valid, flow, odom_t0_t1):
    """liso/tracker/tracking.py:2168-2211 (one batch row) -> fg_kabsch_trafos [K,4,4], bg_kabsch_trafo [4,4],
    st0_T_dyn_motion_warped_bt1 [K,4,4], st1_T_bt1 [K,4,4] (fp64)"""
    K = pos.shape[0]
    fg = np.tile(np.eye(4), (K, 1, 1))
    fg[:, :3, 3] = mean_flow_per_box(pos, dims, rot, pts, valid, flow).astype(np.float64)
    bg = np.linalg.inv(odom_t0_t1)
    # extract_box_motion_transform_without_sensor_odometry (shape_utils.py:583-605)
    s0_T_box0 = box_poses(pos, rot)
    b0_deltaT_b1 = np.linalg.inv(s0_T_box0) @ np.linalg.inv(bg) @ (fg @ s0_T_box0)
    return fg, bg, s0_T_box0 @ b0_deltaT_b1, fg @ s0_T_box0


def match_greedy(iou_matrix, conf, matching_threshold):
    """match_boxes_by_descending_confidence_iou, greedy branch (liso/kabsch/box_groundtruth_matching_iou.py:33-68).
    iou_matrix fp32 [n_gt,n_pred]; conf [n_pred].  The visiting order is argsort(descending) of the confidences (ties in the
    confidences are as ambiguous here as in the reference)."""
    n_true, n_pred = iou_matrix.shape
    order = np.argsort(-conf.reshape(-1), kind="stable")
    return match_greedy_ordered(iou_matrix, order, matching_threshold)


def match_greedy_ordered(iou_matrix, order, matching_threshold):
    n_true, n_pred = iou_matrix.shape
    taken = np.zeros(n_true, dtype=bool)
    pred_mask = np.zeros(n_pred, dtype=bool)
    idx_gt, idx_pred, dists = [], [], []
    thr = np.float32(matching_threshold)
    for p in order:
        best, arg = -np.inf, None
        for g in range(n_true):
            if not taken[g] and iou_matrix[g, p] > best:
                best, arg = iou_matrix[g, p], g
        if best > thr:
            idx_gt.append(arg)
            idx_pred.append(int(p))
            dists.append(best)
            taken[arg] = True
            pred_mask[p] = True
    return (np.array(idx_gt, dtype=np.int64), np.array(idx_pred, dtype=np.int64), np.array(dists, dtype=np.float32), pred_mask,
            taken)
