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


class SlimTrainer:
    """SLIM self-supervised train step, mirror of liso/slim/experiment.py:834-919 (`train_one_step`) with the optimizer /
    schedule factory of :200-219: RMSprop(lr 1e-4) + linear warm-up (2000) then linear decay to 5 %; the loss is the
    un-weighted sum over the 6 RAFT iterations.  The reference cloud of every kNN query is bucketed on the device once
    per step (the reference rebuilds a host KD-tree for each of the 12+ queries)."""

    def __init__(self, cfg, device, num_train_samples=1000):
        from liso_amd.slim.model.slim import SLIM
        from liso_amd.utils.learning_rate import get_polynomial_decay_schedule_with_warmup

        self.cfg, self.slim_cfg, self.device = cfg, cfg.SLIM, device
        self.net = SLIM(cfg, num_train_samples=num_train_samples).to(device)
        self.model = self.net
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.model = torch.nn.parallel.DistributedDataParallel(
                self.net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=64,
                broadcast_buffers=False, gradient_as_bucket_view=True)
        assert self.slim_cfg.optimizer == "rmsprop"
        self.optimizer = torch.optim.RMSprop(self.net.parameters(), lr=self.slim_cfg.learning_rate.initial)
        self.lr_scheduler = get_polynomial_decay_schedule_with_warmup(
            optimizer=self.optimizer, num_warmup_steps=self.slim_cfg.learning_rate.warm_up.step_length,
            num_training_steps=self.slim_cfg.iterations.train, lr_end=self.slim_cfg.learning_rate.initial * 0.05)
        half = 0.5 * torch.tensor(cfg.data.bev_range_m).numpy()
        import numpy as np
        self.bev_extent = np.concatenate([-half, half], axis=0)

    def loss(self, sample_t0, sample_t1):
        from liso_amd.slim.slim_loss.knn_graph import KnnIndex
        from liso_amd.slim.slim_loss.slim_loss_adaptor import selfsupervisedSlimSingleScaleLoss

        pc1, m1 = sample_t0["pcl_ta"]["pcl"].to(self.device), sample_t0["pcl_ta"]["pcl_is_valid"].to(self.device)
        pc2, m2 = sample_t1["pcl_ta"]["pcl"].to(self.device), sample_t1["pcl_ta"]["pcl_is_valid"].to(self.device)
        ext = [float(v) for v in self.bev_extent]
        # bucket both clouds before the network runs: the `all valid` test is the step's only device->host sync and
        # costs nothing while the queue is still empty
        idx1 = [KnnIndex(pc1[b][:, :3], extent=ext) for b in range(pc1.shape[0])] if bool(m1.all()) else None
        idx2 = [KnnIndex(pc2[b][:, :3], extent=ext) for b in range(pc2.shape[0])] if bool(m2.all()) else None
        preds_fw, preds_bw = self.model(sample_t0, sample_t1, None)
        total = torch.zeros(1, device=self.device)
        for pfw, pbw in zip(preds_fw, preds_bw):
            total = total + selfsupervisedSlimSingleScaleLoss(
                pc1=pc1, valid_mask_pc1=m1, pc2=pc2, valid_mask_pc2=m2, pred_fw=pfw, pred_bw=pbw,
                moving_thresh_module=self.net.moving_dynamicness_threshold, loss_cfg=self.slim_cfg.losses.unsupervised,
                model_cfg=self.slim_cfg.model, bev_extent=self.bev_extent, metrics_collector={},
                knn_index_pc1=idx1, knn_index_pc2=idx2)
        return total, preds_fw, preds_bw

    def step(self, sample_t0, sample_t1):
        self.model.train()
        total, _, _ = self.loss(sample_t0, sample_t1)
        self.optimizer.zero_grad(set_to_none=True)
        total.backward()
        self.optimizer.step()
        self.lr_scheduler.step()
        return total.detach()


class LisoLoopTrainer:
    """One fused LISO iteration per sample pair: SLIM forward (no_grad) -> per-point flow ->
    FlowClusterDetector (BEV dynamicness, DBSCAN, region moments, z-fit, filters, Kabsch heading/velocity) -> rotated NMS
    (pre 1000 / post 100 / IoU 0.1, liso_config.yml:4,27-28) -> CenterPoint target maps -> detector train step.
    The reference runs these stages as separate jobs that exchange files (flow export: slim/experiment.py:363-471;
    box mining + box DB: tracker/; training: liso_cli.py); here the tensors stay in HBM from the sweep to the gradient.
    Box-DB augmentation and tracking between the stages are outside this loop."""

    def __init__(self, cfg, device, compute_dtype=torch.float32, total_steps=None, slim_state_dict=None):
        from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector
        from liso_amd.slim.model.slim import SLIM

        self.cfg, self.device = cfg, device
        self.slim = SLIM(cfg, num_train_samples=1000).to(device)
        if slim_state_dict is not None:
            self.slim.load_state_dict(slim_state_dict)
        self.slim.eval()
        self.cluster_detector = FlowClusterDetector(cfg).to(device)
        self.detector = DetectorTrainer(cfg, device, compute_dtype=compute_dtype, total_steps=total_steps)
        tc = cfg.data.tracking_cfg
        self.pre_nms, self.post_nms = tc.max_num_boxes_before_nms, tc.max_num_boxes_after_nms
        self.nms_iou = cfg.setdefault("nms_iou_threshold", 0.1)

    @torch.no_grad()
    def mine_boxes(self, sample_t0, sample_t1):
        """-> (Shape [B,K] after NMS, padded with zeros; point flow [B,N,3])"""
        from liso_amd.utils.nms_iou import perform_nms_on_shapes

        preds_fw, _ = self.slim(sample_t0, sample_t1, None)
        flow = preds_fw[-1].aggregated_flow
        sample = dict(sample_t0)
        sample[self.cfg.data.flow_source] = {**sample_t0.get(self.cfg.data.flow_source, {}), "flow_ta_tb": flow}
        boxes = self.cluster_detector(sample, global_step=1)
        if boxes.shape[1] > 0:
            boxes = perform_nms_on_shapes(boxes, max_num_boxes=self.post_nms, overlap_threshold=self.nms_iou,
                                          pre_nms_max_num_boxes=self.pre_nms)
            boxes.set_padding_val_to(0.0)
        return boxes, flow

    def step(self, sample_t0, sample_t1):
        from liso_amd.datasets.targets import render_center_targets

        boxes, _ = self.mine_boxes(sample_t0, sample_t1)
        B = boxes.shape[0]
        if boxes.shape[1] == 0:  # nothing moved: an all-background target (one padded slot)
            z = torch.zeros((B, 1, 3), device=self.device)
            pos, dims, rot, valid = z, z + 1.0, z[..., :1], torch.zeros((B, 1), dtype=torch.bool, device=self.device)
        else:
            pos, dims, rot, valid = boxes.pos.float(), boxes.dims.float().clamp(min=1e-3), boxes.rot.float(), boxes.valid
        out = tuple(int(g) // 4 for g in self.cfg.data.img_grid_size)
        targets = render_center_targets(pos, dims, rot, valid, out, tuple(self.cfg.data.bev_range_m))
        self.last_boxes = boxes
        return self.detector.step(sample_t0["pcl_full_no_ground_ta"], targets)
