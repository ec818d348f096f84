"""Training steps of the hot path.

DetectorTrainer.step mirrors the per-iteration body of liso/kabsch/liso_cli.py:362-618 for the CenterPoint-pillar
network (forward -> centerpoint_loss x cm_loss_weight + rotation regulariser -> backward -> AdamW -> OneCycleLR,
optimizer factory liso_cli.py:792-823).  The reference is single-GPU; data parallelism (one process per GPU,
gradient all-reduce over RCCL/xGMI in a single flat bucket overlapped with backward, per-rank BatchNorm) is the
only addition.
"""
import torch
import torch.distributed as dist

from liso_amd.losses.centerpoint_loss import centerpoint_loss, rotation_vec_on_unit_circle
from liso_amd.networks.simple_net.simple_net import BoxLearner


def get_optimizer_scheduler(cfg, box_predictor, total_steps=None):
    """liso_cli.py:792-823 (train_on_box_source == "gt" branch)"""
    opt = torch.optim.AdamW(box_predictor.parameters(), lr=cfg.optimization.learning_rate, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(
        optimizer=opt, max_lr=cfg.optimization.learning_rate, pct_start=0.4, base_momentum=0.85, max_momentum=0.95,
        div_factor=10.0, total_steps=(total_steps or cfg.optimization.num_training_steps) + 2)
    return opt, sched


class DetectorTrainer:
    def __init__(self, cfg, device, compute_dtype=torch.float32, total_steps=None):
        self.cfg, self.device = cfg, device
        self.net = BoxLearner(cfg).to(device)
        self.net.model.set_compute_dtype(compute_dtype)
        if compute_dtype != torch.float32:
            self.net.model.rpn.to(memory_format=torch.channels_last)
            self.net.model.center_head.to(memory_format=torch.channels_last)
        self.model = self.net
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # ~19 MB of fp32 gradients: one flat bucket, launched as backward reaches the first layer's grads
            self.model = torch.nn.parallel.DistributedDataParallel(
                self.net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=64,
                broadcast_buffers=False, gradient_as_bucket_view=True)
        self.optimizer, self.lr_scheduler = get_optimizer_scheduler(cfg, self.net, total_steps)

    def loss(self, pcls, targets):
        """liso_cli.py:452-614"""
        cfg = self.cfg
        pred_boxes, decoded, activated, aux = self.model(None, pcls, None, centermaps_gt=None)
        sup = cfg.loss.supervised.supervised_on_clusters
        gt_maps = {a: targets[a] for a in sup.attrs}
        mask = targets["center_bool_mask"]
        ignore = targets.get("ignore_region_is_true_mask", torch.zeros_like(mask))
        losses = centerpoint_loss(loss_cfg=cfg.loss, raw_activated_pred_box_maps=activated, decoded_pred_box_maps=decoded,
                                  gt_maps=gt_maps, gt_center_mask=mask,
                                  rotation_loss_weights_map=torch.ones_like(gt_maps["probs"]),
                                  box_prediction_cfg=cfg.box_prediction, ignore_region_is_true_mask=ignore)
        total = 0.0
        for v in losses.values():
            total = total + sup.weight * v
        rr = cfg.box_prediction.rotation_representation
        if rr.method == "vector":  # main_utils.py:119-134
            total = total + rotation_vec_on_unit_circle(activated) * rr.regul_weight
        return total, losses, pred_boxes

    def step(self, pcls, targets):
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        total, losses, _ = self.loss(pcls, targets)
        total.backward()
        self.optimizer.step()
        self.lr_scheduler.step()
        return total.detach()
