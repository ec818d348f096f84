"""Generates tests/golden/train_step_reference.npz: the captured reference steps SURVEY.md 8a row F1 asks for.

(1) Optimizer / scheduler factories, as trajectories on a small dummy parameter set driven by a fixed gradient sequence:
      liso.kabsch.liso_cli.get_optimizer_scheduler            (liso_cli.py:792-823, AdamW + OneCycleLR, "gt" and "mined")
        -- the FunctionDef node is compiled from the reference file where it lies (the module as a whole imports the
           whole training stack); nothing of it is stored.
      torch.optim.RMSprop + liso.utils.learning_rate.get_polynomial_decay_schedule_with_warmup, called with the argument
      expressions of liso/slim/experiment.py:200-219 (RMSprop(lr=initial); warm-up `step_length`, `iterations.train`,
      lr_end = initial * 0.05).
(2) Two consecutive detector train steps of the CenterPoint-pillar network (liso_cli.py:452-618) assembled from the
    reference's own python: mmdet3d voxel_generator + PillarFeatureNet + PointPillarsScatter glued as
    pcl_to_feature_grid.py:58-102, liso.networks.centerpoint.{rpn.RPN, center_head.CenterHead} configured as
    centerpoint_net.py:22-65, activations simple_net_utils.py:8-14, liso.kabsch.output_modification.output_modification,
    liso.losses.centerpoint_loss.centerpoint_loss, main_utils.rotation_vec_on_unit_circle, the factory of (1).
    Full channel widths (4.8 M parameters); initial weights come from tests/golden/keyed_init.py (a function of the key
    names), inputs are two small synthetic clouds on a 64 x 64 grid; the fixture holds losses, per-parameter gradient
    norms, sampled post-step weights and learning rates.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_train_step_golden.py
"""
import ast
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_targets_golden import cfg as mkcfg, import_with_stubs  # noqa: E402  (puts /root/reference on sys.path)

import ref_import  # noqa: E402
from keyed_init import keyed_state_dict, sample_indices  # noqa: E402
from oracle import pillars as OP  # noqa: E402  (synthetic cloud generator + pillar geometry only)

GRID, RANGE, ZCUT = 64, 40.0, 10.0


def function_from_reference_file(path, name, env):
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), env)
    return env[name]


def dummy_problem():
    g = torch.Generator().manual_seed(0)
    params = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g), torch.randn(3, 2, 2, generator=g)]
    grads = [[torch.randn(p.shape, generator=g) * (0.1 + 0.3 * s) for p in params] for s in range(14)]
    return params, grads


def run_trajectory(make, params, grads):
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt, sched = make(ps)
    traj, lrs = [], []
    for gs in grads:
        for p, g_ in zip(ps, gs):
            p.grad = g_.clone()
        opt.step()
        sched.step()
        traj.append(torch.cat([p.detach().reshape(-1) for p in ps]).numpy().copy())
        lrs.append(opt.param_groups[0]["lr"])
    return np.stack(traj), np.array(lrs)


def main():
    out = {}
    # ---- (1) factories --------------------------------------------------------------------------------------------------
    get_opt = function_from_reference_file("/root/reference/liso/kabsch/liso_cli.py", "get_optimizer_scheduler", {"torch": torch})
    from liso.utils.learning_rate import get_polynomial_decay_schedule_with_warmup

    params, grads = dummy_problem()

    class _Net:
        def __init__(self, ps):
            self.ps = ps

        def parameters(self):
            return self.ps

    c_gt = mkcfg({"optimization": {"learning_rate": 1e-3, "num_training_steps": 12}, "data": {"train_on_box_source": "gt"}})
    c_mined = mkcfg({"optimization": {"learning_rate": 1e-3, "num_training_steps": 12,
                                      "rounds": {"active": True, "steps_per_round": 6, "drop_net_weights_every_nth_round": 2}},
                     "data": {"train_on_box_source": "mined"}})
    out["adamw_gt_traj"], out["adamw_gt_lr"] = run_trajectory(lambda ps: get_opt(c_gt, _Net(ps)), params, grads[:13])
    out["adamw_mined_traj"], out["adamw_mined_lr"] = run_trajectory(lambda ps: get_opt(c_mined, _Net(ps)), params, grads[:13])
    slim = mkcfg({"learning_rate": {"initial": 1e-4, "warm_up": {"step_length": 5}}, "iterations": {"train": 12}})

    def make_slim(ps):  # experiment.py:200-219
        opt = torch.optim.RMSprop(ps, lr=slim.learning_rate.initial)
        return opt, get_polynomial_decay_schedule_with_warmup(optimizer=opt, num_warmup_steps=slim.learning_rate.warm_up.step_length,
                                                             num_training_steps=slim.iterations.train,
                                                             lr_end=slim.learning_rate.initial * 0.05)

    out["rmsprop_traj"], out["rmsprop_lr"] = run_trajectory(make_slim, params, grads)

    # ---- (2) two detector train steps ------------------------------------------------------------------------------------
    def _imp():
        from liso.kabsch.output_modification import output_modification
        from liso.losses.centerpoint_loss import centerpoint_loss
        from liso.networks.centerpoint.center_head import CenterHead
        from liso.networks.centerpoint.rpn import RPN
        from liso.utils.bev_utils import get_metric_voxel_center_coords
        return output_modification, centerpoint_loss, CenterHead, RPN, get_metric_voxel_center_coords

    output_modification, centerpoint_loss, CenterHead, RPN, get_centers = import_with_stubs(_imp)
    # main_utils as a module imports the evaluation stack (tensorboard, ...): take the one function from the file
    rot_reg = function_from_reference_file("/root/reference/liso/kabsch/main_utils.py", "rotation_vec_on_unit_circle", {"torch": torch})
    utils, enc, sc, vg = ref_import.load_mmdet3d_pillar_modules()
    pc_range, voxel_size = OP.pillar_geometry((RANGE, RANGE), (GRID, GRID), ZCUT)
    pfn = enc.PillarFeatureNet(in_channels=4, feat_channels=[64], with_distance=False, voxel_size=voxel_size,
                               norm_cfg={"type": "BN1d", "eps": 0.001, "momentum": 0.01}, point_cloud_range=pc_range)
    norm = {"affine": True, "track_running_stats": True}                       # liso_config.yml:191-194
    rpn = RPN(layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[64, 128, 256], us_layer_strides=[0.5, 1, 2],
              us_num_filters=[128, 128, 128], num_input_features=64, norm_cfg=norm)   # centerpoint_net.py:22-59
    heads = {"pos": (3, 2), "dims": (3, 2), "rot": (2, 2), "probs": (1, 2)}
    head = CenterHead(common_heads=heads, norm_cfg=norm, in_channels=384, stride=1, share_conv_channel=64)
    modules = {"model.pfn.pts_voxel_encoder.": pfn, "model.rpn.": rpn, "model.center_head.": head}
    shapes = {pre + k: (tuple(v.shape), v.dtype) for pre, m in modules.items() for k, v in m.state_dict().items()}
    init = keyed_state_dict(shapes)
    for pre, m in modules.items():
        m.load_state_dict({k[len(pre):]: v for k, v in init.items() if k.startswith(pre)}, strict=True)
        m.train()
    scatter = sc.PointPillarsScatter(in_channels=64, output_shape=(GRID, GRID))
    pcls = [OP.synthetic_cloud(6000, 40 + i, RANGE, 4) for i in range(2)]
    out["pcl_0"], out["pcl_1"] = pcls
    out["key_order"] = np.array([k for k in shapes])

    class _P(torch.nn.Module):  # parameter container in the order of BoxLearner.parameters(): pfn, rpn, center_head
        def __init__(self):
            super().__init__()
            self.pfn, self.rpn, self.center_head = pfn, rpn, head

    net = _P()
    c_det = mkcfg({"optimization": {"learning_rate": 1e-3, "num_training_steps": 8}, "data": {"train_on_box_source": "gt"}})
    opt, sched = get_opt(c_det, net)
    box_cfg = mkcfg({"position_representation": {"method": "local_relative_offset", "num_box_pos_dims": 3,
                                                 "box_z_pos_prior_min": -1.5, "box_z_pos_prior_max": -0.5},
                     "rotation_representation": {"method": "vector", "norm_vector_len": False},
                     "dimensions_representation": {"method": "predict_abs_size"},
                     "activations": {"pos": "tanh", "dims": "softplus", "rot": "none", "probs": "none"}})
    data_cfg = mkcfg({"bev_range_m": (RANGE, RANGE)})
    loss_cfg = mkcfg({"supervised": {"centermaps": {"confidence_target": "gaussian"}}})
    H = GRID // 4
    centers = torch.from_numpy(get_centers(RANGE, RANGE, np.array([H, H])).astype(np.float32)[..., :2])
    g = torch.Generator().manual_seed(5)
    gt = {"probs": torch.rand(2, H, H, 1, generator=g) ** 4, "rot": torch.randn(2, H, H, 2, generator=g),
          "dims": torch.rand(2, H, H, 3, generator=g) * 4 + 0.5, "pos": torch.randn(2, H, H, 3, generator=g) * 8}
    center_mask = torch.rand(2, H, H, generator=g) > 0.95
    gt["probs"][center_mask] = 1.0
    for k, v in gt.items():
        out["gt_" + k] = v.numpy()
    out["center_mask"] = center_mask.numpy()

    def forward_loss():
        voxels, coors, nums = [], [], []
        for b, p in enumerate(pcls):  # pcl_to_feature_grid.py:58-84
            v, c, n = vg.points_to_voxel(p, voxel_size, pc_range, max_points=20, reverse_index=True, max_voxels=40000)
            c = c[:, [0, 2, 1]]
            coors.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
            voxels.append(v), nums.append(n)
        vt, ct, nt = torch.from_numpy(np.concatenate(voxels)), torch.from_numpy(np.concatenate(coors)), torch.from_numpy(np.concatenate(nums))
        bev = scatter(pfn(vt, nt, ct), ct, len(pcls))                              # :86-102
        pred = head(rpn(bev))
        raw = {k: v.permute(0, 2, 3, 1) for k, v in pred.items()}                 # centerpoint_net.py:111
        act = {"pos": torch.tanh(raw["pos"]), "dims": torch.nn.functional.softplus(raw["dims"]), "rot": raw["rot"],
               "probs": raw["probs"]}                                              # simple_net.py:118-121
        dec = output_modification({k: v.clone() for k, v in act.items()}, box_cfg, data_cfg, "boxes", centers)
        losses = centerpoint_loss(loss_cfg=loss_cfg, decoded_pred_box_maps=dec, raw_activated_pred_box_maps=act, gt_maps=gt,
                                  gt_center_mask=center_mask, rotation_loss_weights_map=torch.ones_like(gt["probs"]),
                                  box_prediction_cfg=box_cfg, ignore_region_is_true_mask=torch.zeros_like(center_mask))
        total = 0.0
        for v in losses.values():
            total = total + 1.0 * v                                                # liso_cli.py:469,588
        return total + rot_reg(act) * 0.0001, raw                                  # main_utils.py:119-134

    named = {pre + k: p for pre, m in modules.items() for k, p in m.named_parameters()}
    for step in range(2):
        opt.zero_grad()
        total, raw = forward_loss()
        total.backward()
        out[f"step{step}_loss"] = np.array(float(total))
        out[f"step{step}_raw_probs"] = raw["probs"].detach().numpy()
        out[f"step{step}_grad_norms"] = np.array([float(p.grad.norm()) if p.grad is not None else -1.0 for p in named.values()])
        opt.step()
        sched.step()
        out[f"step{step}_lr"] = np.array(opt.param_groups[0]["lr"])
        out[f"step{step}_weight_norms"] = np.array([float(p.detach().norm()) for p in named.values()])
        out[f"step{step}_weight_samples"] = np.stack(
            [np.pad(p.detach().reshape(-1)[sample_indices(k, p.numel())].numpy(), (0, 64 - min(64, p.numel()))) for k, p in named.items()])
    out["param_keys"] = np.array(list(named))
    out["bn_running_mean_rpn_blocks_0_2"] = rpn.blocks[0][2].running_mean.numpy().copy()
    out["bn_running_var_head_shared_1"] = head.shared_conv[1].running_var.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "train_step_reference.npz"), **out)
    print("losses", float(out["step0_loss"]), float(out["step1_loss"]), "lr", float(out["step0_lr"]), float(out["step1_lr"]),
          "n params", len(named), "size", os.path.getsize(os.path.join(HERE, "train_step_reference.npz")))


if __name__ == "__main__":
    main()
