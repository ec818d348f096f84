"""Minimal attribute-access config (the reference uses OmegaConf, config_helper/config_helper/config.py; only
`.a.b` access and `.setdefault` are needed by the hot path, pcl_to_feature_grid.py:14,37) plus the hot-path
defaults copied as *values* from liso/config/liso_config.yml (line numbers cited per key)."""
import copy


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return to_attr(copy.deepcopy(dict(self), memo))


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    return d


def apply_slim_simple_knn_training(cfg):
    """the `slim_simple_knn_training` + `slim_set_cls_output_all_static` overlays (liso_config.yml:722-731, :788-795)"""
    u = cfg.SLIM.losses.unsupervised
    u.opposite_flow_penalty_factor = 0.0
    u.fw_bw_static_trafo_penalty_factor = 0.0
    u.static_flow_penalty_factor = 0.0
    u.knn_loss_penalty_factor = 1.0
    om = cfg.SLIM.model.output_modification
    om.static_logit, om.dynamic_logit, om.ground_logit, om.dynamic_flow = True, False, False, "zero"
    return cfg


def default_cfg(grid=512, bev_range_m=100.0, use_lidar_intensity=True):
    """CenterPoint-pillar detector + SLIM settings of the reference's KITTI/nuScenes overlays."""
    return to_attr({
        "data": {
            "shapes": {"name": "boxes"},
            "use_ground_for_network": False,
            "bev_range_m": (bev_range_m, bev_range_m),      # liso_config.yml:515-522
            "img_grid_size": (grid, grid),
            "z_pillar_cutoff_value": 10.0,                  # :116
            "use_lidar_intensity": use_lidar_intensity,     # :47
            "limit_pillar_height": True,                    # :115
            "pillar_height_range_m": (-2.0, 1.0),           # :117-119
            "flow_source": "gt", "odom_source": "gt",       # :111-112
            "tracking_cfg": {"min_points_in_box": 20, "max_num_boxes_after_nms": 100, "max_num_boxes_before_nms": 1000,  # :21-34
                             "flow_cluster_detector_min_obj_speed_mps": 1.0},
        },
        "network": {
            "name": "centerpoint",
            "centerpoint": {
                "reduce_receptive_field": 0, "hid_dim": 64, "use_baseline_parameters": True,  # :187-190
                "channel_reduction_factor": 1,
                "batch_norm": {"kwargs": {"affine": True, "track_running_stats": True}},      # :191-194
            },
        },
        "box_prediction": {
            "position_representation": {"method": "local_relative_offset", "num_box_pos_dims": 3,
                                        "box_z_pos_prior_min": -1.5, "box_z_pos_prior_max": -0.5},  # :200-201
            "rotation_representation": {"method": "vector", "norm_vector_len": False,         # :702-705, :206
                                        "regularization": "rot_vec_on_unit_circle", "regul_weight": 0.0001},
            "dimensions_representation": {"method": "predict_abs_size"},                      # :695-697
            "activations": {"pos": "tanh", "dims": "softplus", "rot": "none", "probs": "none"},  # :617-631
            "output_modification": {"pos": "none", "dims": "none", "rot": "none", "probs": "none"},
        },
        "loss": {"supervised": {"centermaps": {"active": True, "confidence_target": "gaussian"},   # :179
                                "supervised_on_clusters": {"active": True, "weight": 1.0,
                                                           "attrs": ("pos", "dims", "rot", "probs")}}},  # :155-162
        "SLIM": {                                                                                     # :231-330
            "optimizer": "rmsprop", "batch_size": 1,
            "phases": {"train": {"mode": "unsupervised"}},
            "iterations": {"train": 150000},                                                          # :240
            "learning_rate": {"initial": 0.0001, "warm_up": {"initial": 0.01, "step_length": 2000}},   # :251-258
            "losses": {"unsupervised": {                                                               # :259-287
                "fw_bw_static_trafo_penalty_factor": 1.0, "knn_loss_penalty_factor": 1.0,
                "artificial_labels": {"use_static_aggr_flow": True, "cross_entropy_penalty": 0.0, "weight_mode": "constant",
                                      "gauss_widths": None, "knn_mode": "point"},
                "knn_on_dynamic_penalty": 0.0, "knn_on_static_penalty": 0.0, "knn_dist_measure": "point",
                "knn_loss": {"L1_delta": 0.0, "drop_outliers__perc": 0.0, "fov_mode": "mask_close_fov",
                             "range_based_weights": {"slope_sign": -1.0, "weight_slope": 0.0, "weight_at_range_0": 0.0,
                                                     "max_weight_clip_at": 100.0, "min_weight_clip_at": 1.0}},
                "opposite_flow_penalty_factor": 0.0, "static_flow_penalty_factor": 1.0,
                "temporal_cls_consistency_penalty_factor": 0.0, "use_epsilon_for_weighted_pc_alignment": False,
            }},
            "model": {
                "name": "raft", "dropout_rate": 0, "raft_fnet_norm": "instance_affine",              # :290-292
                "feature_downsampling_factor": 8, "num_iters": 6, "num_pred_iters": 6,                 # :293-296
                "flow_maps_archi": "single",                                                           # :297
                "corr_cfg": {"module": "all", "sampler": "bilinear", "search_radius": 3, "num_levels": 4},  # :298-302
                "predict_weight_for_static_aggregation": False, "use_static_aggr_flow_for_aggr_flow": False,  # :311-312
                "dynamic_flow_is_non_rigid_flow": False,
                "output_modification": {"disappearing_logit": False, "static_logit": "net", "dynamic_logit": "net",   # :303-310
                                        "ground_logit": False, "dynamic_flow": "net", "static_flow": "net",
                                        "dynamic_flow_grad_scale": 1.0},
                "point_pillars": {"nbr_point_feats": 64}, "u_net": {"final_scale": 1},
            },
        },
        "mask_rendering": {"softness_fun": "cauchy", "pred_sigmoid_slope": 15.0, "obj_dim_scale_buffer": 0.25},  # :130-134
        "svd_backend": "symm_ortho",                                                                  # :230
        "optimization": {"learning_rate": 0.001, "num_training_steps": 350000},                      # :137-138
    })
