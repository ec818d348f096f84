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


def default_cfg(grid=512, bev_range_m=100.0, use_lidar_intensity=True):
    """CenterPoint-pillar detector + SLIM settings of the reference's KITTI/nuScenes overlays."""
    return to_attr({
        "data": {
            "shapes": {"name": "boxes"},
            "bev_range_m": (bev_range_m, bev_range_m),      # liso_config.yml:515-522
            "img_grid_size": (grid, grid),
            "z_pillar_cutoff_value": 10.0,                  # :116
            "use_lidar_intensity": use_lidar_intensity,     # :47
            "limit_pillar_height": True,                    # :115
            "pillar_height_range_m": (-2.0, 1.0),           # :117-119
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
            "model": {
                "name": "raft", "dropout_rate": 0, "raft_fnet_norm": "instance_affine",              # :290-292
                "feature_downsampling_factor": 8, "num_iters": 6, "num_pred_iters": 6,                 # :293-296
                "flow_maps_archi": "single",                                                           # :297
                "corr_cfg": {"module": "all", "sampler": "bilinear", "search_radius": 3, "num_levels": 4},  # :298-302
                "predict_weight_for_static_aggregation": False, "use_static_aggr_flow_for_aggr_flow": False,  # :311-312
                "dynamic_flow_is_non_rigid_flow": False,
                "point_pillars": {"nbr_point_feats": 64}, "u_net": {"final_scale": 1},
            },
        },
        "mask_rendering": {"softness_fun": "cauchy", "pred_sigmoid_slope": 15.0, "obj_dim_scale_buffer": 0.25},  # :130-134
        "svd_backend": "symm_ortho",                                                                  # :230
        "optimization": {"learning_rate": 0.001, "num_training_steps": 350000},                      # :137-138
    })
