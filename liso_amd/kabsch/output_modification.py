"""Box decode (network convention -> metric boxes).  Mirror of liso/kabsch/output_modification.py."""
import torch


def modify_pred_pos(pred_pos, box_pred_cfg, data_cfg, pillar_center_coors_m):
    """reference :4-45"""
    method = box_pred_cfg.position_representation.method
    if method == "global_relative":
        assert box_pred_cfg.activations.pos in ("tanh",)
        return pred_pos * torch.tensor(data_cfg.bev_range_m, device=pred_pos.device) * 0.6
    if method == "local_relative_offset":
        assert box_pred_cfg.activations.pos in ("tanh", "none"), box_pred_cfg.activations.pos
        assert len(pred_pos.shape) == 4, pred_pos.shape
        assert pillar_center_coors_m.shape[:-1] == pred_pos.shape[1:3]
        res = (torch.tensor(data_cfg.bev_range_m) / torch.tensor(pred_pos.shape[1:3])).to(pred_pos.device)
        out = pillar_center_coors_m[None, ...] + res * 0.5 * pred_pos[..., :2]
        pr = box_pred_cfg.position_representation
        if pr.num_box_pos_dims == 3:
            assert pred_pos.shape[-1] == 3, pred_pos.shape
            z = pr.box_z_pos_prior_min + 0.5 * (pred_pos[..., -1:] + 1.0) * (pr.box_z_pos_prior_max - pr.box_z_pos_prior_min)
            out = torch.cat([out, z], dim=-1)
        return out
    if method == "global_absolute":
        return pred_pos
    raise NotImplementedError(method)


def maybe_flatten_anchors_except_for(box_vars_pred, do_not_flatten=("pos",)):
    """reference :47-55"""
    for k, v in box_vars_pred.items():
        if k not in do_not_flatten and len(v.shape) == 4:
            box_vars_pred[k] = torch.flatten(v, start_dim=1, end_dim=2)
    return box_vars_pred


def box_pred_convention_to_gt_convention(box_vars_pred, box_pred_cfg, data_cfg, pillar_center_coors_m):
    """reference :58-127"""
    dm = box_pred_cfg.dimensions_representation.method
    if dm == "predict_aspect_ratio":
        scale, ar_inv = torch.split(box_vars_pred["dims"], 1, dim=-1)
        d = box_pred_cfg.dimensions_representation
        length = d.box_len_prior_min + scale * (d.box_len_prior_max - d.box_len_prior_min)
        box_vars_pred["dims"] = torch.cat([length, length * ar_inv], dim=-1)
    elif dm == "predict_log_size":
        box_vars_pred["dims"] = torch.exp(box_vars_pred["dims"])
    elif dm != "predict_abs_size":
        raise NotImplementedError(dm)
    rm = box_pred_cfg.rotation_representation.method
    if rm == "vector":
        vec = box_vars_pred["rot"]
        if box_pred_cfg.rotation_representation.norm_vector_len:
            vec = torch.nn.functional.normalize(vec, p=2.0, dim=-1)
        sin_yaw, cos_yaw = torch.split(vec, 1, dim=-1)  # rot[0]=sin "y", rot[1]=cos "x" (:91-101)
        box_vars_pred["rot"] = torch.atan2(sin_yaw, cos_yaw)
    elif rm == "class_bins":
        box_vars_pred["rot"] = torch.argmax(box_vars_pred["rot"], dim=-1, keepdim=True) * (2 * torch.pi / 36)
    elif rm != "direct":
        raise NotImplementedError(rm)
    box_vars_pred["pos"] = modify_pred_pos(box_vars_pred["pos"], box_pred_cfg, data_cfg, pillar_center_coors_m)
    return box_vars_pred


def output_modification(box_vars_pred, box_pred_cfg, data_cfg, shape_name, pillar_center_coors_m):
    """reference :130-148"""
    box_vars_pred = {k: v.clone() for k, v in box_vars_pred.items()}
    if shape_name != "boxes":
        raise NotImplementedError(shape_name)
    return box_pred_convention_to_gt_convention(box_vars_pred, box_pred_cfg, data_cfg, pillar_center_coors_m)
