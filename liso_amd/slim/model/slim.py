"""SLIM = RAFT on BEV pillars + per-direction HeadDecoder.  Mirror of liso/slim/model/slim.py:10-156 (same
constructor, sub-module names `head_decoder_fw/bw`, `moving_dynamicness_threshold`, `raft_network`, same forward
signature and outputs).  The hard-coded `"cuda"` placements of the reference (:56,:62) become "the module's device"."""
import numpy as np
import torch
from torch import nn

from liso_amd.slim.model.head_decoder import HeadDecoder
from liso_amd.slim.model.raft_mod import RAFT
from liso_amd.slim.slim_loss.movavg_cls_threshold import MovingAverageThreshold
from liso_amd.slim.slim_loss.static_aggregation import BevGatherPlan


def get_network_input_pcls(cfg, sample_data, time_key: str, to_device=None):
    """liso/kabsch/main_utils.py:247-261"""
    key = f"pcl_full_w_ground_{time_key}" if cfg.data.use_ground_for_network else f"pcl_full_no_ground_{time_key}"
    return [el.to(to_device) for el in sample_data[key]] if to_device else sample_data[key]


def _slice_prediction(pred, sl):
    """the samples `sl` of a batched HeadDecoder result (tensors with a leading batch dimension are sliced, 0-d tensors and
    other values are shared)"""
    out = type(pred)()
    for k, v in pred.items():
        if isinstance(v, dict):
            out[k] = _slice_prediction(v, sl)
        elif torch.is_tensor(v) and v.ndim >= 1:
            out[k] = v[sl]
        else:
            out[k] = v
    return out


class SLIM(nn.Module):
    def __init__(self, cfg, num_train_samples, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.cfg, self.slim_cfg = cfg, cfg.SLIM
        half = 0.5 * np.array(cfg.data.bev_range_m)
        bev_pc_range = np.concatenate([-half, half], axis=0)
        self.head_decoder_fw = HeadDecoder(self.slim_cfg, bev_extent=bev_pc_range, name="head_decoder_forward")
        self.head_decoder_bw = HeadDecoder(self.slim_cfg, bev_extent=bev_pc_range, name="head_decoder_backward")
        assert self.slim_cfg.phases.train.mode in ["supervised", "unsupervised"]
        if self.slim_cfg.phases.train.mode == "unsupervised":
            num_still = None
        else:
            num_still = self.slim_cfg.data.train.num_still_points * self.slim_cfg.model.num_iters
        self.moving_dynamicness_threshold = MovingAverageThreshold(num_train_samples=num_train_samples,
                                                                   num_moving=621013971, num_still=num_still)
        self.raft_network = RAFT(cfg=cfg, head_decoder_fw=self.head_decoder_fw, head_decoder_bw=self.head_decoder_bw)

    @torch.no_grad()
    def infer_point_flow_t0_t1(self, sample_data_t0, sample_data_t1, canvases=None, dynamicness_threshold=None):
        """Per-point flow t0 -> t1 of the sweep at t0 (`aggregated_flow` of the last RAFT iteration, [B,N,3]): what
        FlowClusterDetector consumes.  One flow direction, one decode -- the training forward produces 12.
        `dynamicness_threshold`: `moving_dynamicness_threshold.value()` computed by the caller (a device scan: callers that
        replay this method from a hipGraph evaluate it eagerly, liso_amd/utils/graph_safety.py)."""
        dev = next(self.raft_network.parameters()).device
        if canvases is not None:  # (the caller ran the pillar encoder: the network-input clouds are not read here)
            net_out, aux = self.raft_network.infer_forward_direction(None, None, canvases=canvases)
        else:
            net_out, aux = self.raft_network.infer_forward_direction(
                get_network_input_pcls(self.cfg, sample_data_t0, "ta", to_device=dev),
                get_network_input_pcls(self.cfg, sample_data_t1, "ta", to_device=dev))
        pa = sample_data_t0["pcl_ta"]
        pred = self.head_decoder_fw(
            net_out, self.moving_dynamicness_threshold.value() if dynamicness_threshold is None else dynamicness_threshold,
            pc=pa["pcl"].to(dev),
            pointwise_voxel_coordinates=pa["pillar_coors"].to(dev), pointwise_valid_mask=pa["pcl_is_valid"].to(dev),
            filled_pillar_mask=torch.squeeze(aux["t0"]["bev_net_input_dbg"] > 0.5, dim=1),
            odom=sample_data_t0["gt"]["odom_ta_tb"].to(dev), inv_odom=sample_data_t1["gt"]["odom_ta_tb"].to(dev), summaries=None,
            dynamic_flow_is_non_rigid_flow=self.slim_cfg.model.dynamic_flow_is_non_rigid_flow, pointwise_only=True,
            aggregated_flow_only=True)
        return pred.aggregated_flow

    @torch.no_grad()
    def infer_export_predictions(self, sample_data_t0, sample_data_t1):
        """What the flow export consumes (liso/slim/experiment.py:363-471: `bev_raw_flow_t0_t1`, `bev_raw_flow_t1_t0`,
        `bev_dynamicness_*` of the LAST RAFT iteration): both flow directions in one batched network pass, one dense decode per
        direction -> (pred_fw, pred_bw) with `.modified_network_output.{static_flow, dynamicness}` like forward()'s preds[-1];
        `liso_amd.slim.flow_io.flow_export_dict([pred_fw], [pred_bw], threshold, bev_range_m)` turns them into the file's arrays.
        The training forward decodes 12 outputs for the same two."""
        dev = next(self.raft_network.parameters()).device
        net_fw, net_bw, aux = self.raft_network.infer_both_directions(
            get_network_input_pcls(self.cfg, sample_data_t0, "ta", to_device=dev),
            get_network_input_pcls(self.cfg, sample_data_t1, "ta", to_device=dev))
        thr = self.moving_dynamicness_threshold.value()
        common = dict(dynamicness_threshold=thr, summaries=None, gt_flow_bev=None, ohe_gt_stat_dyn_ground_label_bev_map=None,
                      dynamic_flow_is_non_rigid_flow=self.slim_cfg.model.dynamic_flow_is_non_rigid_flow)
        preds = []
        for net, dec, sa, sb, occ in ((net_fw, self.head_decoder_fw, sample_data_t0, sample_data_t1, aux["t0"]["bev_net_input_dbg"]),
                                      (net_bw, self.head_decoder_bw, sample_data_t1, sample_data_t0, aux["t1"]["bev_net_input_dbg"])):
            pa = sa["pcl_ta"]
            preds.append(dec(net, pointwise_valid_mask=pa["pcl_is_valid"].to(dev), pointwise_voxel_coordinates=pa["pillar_coors"].to(dev),
                             pc=pa["pcl"].to(dev), filled_pillar_mask=torch.squeeze(occ > 0.5, dim=1),
                             odom=sa["gt"]["odom_ta_tb"].to(dev), inv_odom=sb["gt"]["odom_ta_tb"].to(dev), **common))
        return preds[0], preds[1]

    def build_gather_plan(self, sample_data_t0, sample_data_t1, n_it, grid_hw):
        """point -> BEV cell lists of the batch the decoder sees in forward(): [forward samples x n_it | backward samples x n_it].
        Depends on the sweeps only (a large device sort): callers that replay forward() from a hipGraph build it eagerly."""
        dev = next(self.raft_network.parameters()).device
        pa, pb = sample_data_t0["pcl_ta"], sample_data_t1["pcl_ta"]
        B = pa["pcl"].shape[0]
        fs = self.slim_cfg.model.u_net.final_scale
        cat = lambda a, b: torch.cat([a.to(dev), b.to(dev)], dim=0)  # noqa: E731
        valid, coors = cat(pa["pcl_is_valid"], pb["pcl_is_valid"]), cat(pa["pillar_coors"], pb["pillar_coors"])
        return BevGatherPlan.tiled(torch.div(coors, fs, rounding_mode="trunc"), valid, grid_hw, n_it, B).prepare_backward()

    def forward(self, sample_data_t0, sample_data_t1, summaries=None, canvases=None, gather_plan=None):
        """`canvases` (extension): the pillar canvases of both sweeps, `raft_network.encode_pillars(...)`, computed by the caller;
        `gather_plan` (extension): `build_gather_plan(...)` of these samples, computed by the caller"""
        dev = next(self.raft_network.parameters()).device
        if canvases is not None:
            out_fw, out_bw, aux = self.raft_network(None, None, canvases=canvases)
        else:
            out_fw, out_bw, aux = self.raft_network(get_network_input_pcls(self.cfg, sample_data_t0, "ta", to_device=dev),
                                                    get_network_input_pcls(self.cfg, sample_data_t1, "ta", to_device=dev))
        filled0 = torch.squeeze(aux["t0"]["bev_net_input_dbg"] > 0.5, dim=1)
        filled1 = torch.squeeze(aux["t1"]["bev_net_input_dbg"] > 0.5, dim=1)
        thr = self.moving_dynamicness_threshold.value()
        nri = self.slim_cfg.model.dynamic_flow_is_non_rigid_flow
        preds_fw, preds_bw = [], []
        self.stacked_predictions = None
        fs = self.slim_cfg.model.u_net.final_scale
        pa, pb = sample_data_t0["pcl_ta"], sample_data_t1["pcl_ta"]
        common = dict(dynamicness_threshold=thr, summaries=summaries, gt_flow_bev=None, ohe_gt_stat_dyn_ground_label_bev_map=None,
                      dynamic_flow_is_non_rigid_flow=nri)
        if pa["pcl"].shape == pb["pcl"].shape and getattr(self, "batch_decoding", True):
            # both directions through ONE decoder call per RAFT iteration (batch = [forward samples; backward samples]):
            # the decoder has no parameters and treats samples independently (its only batch-wide quantities are the global
            # logit extrema of the `True`/`False` output modes, head_decoder.py:779-955, which the reference also takes over
            # whatever batch it is given; they only offset logits by +-100 and leave the class probabilities unchanged)
            B = pa["pcl"].shape[0]
            cat = lambda a, b: torch.cat([a.to(dev), b.to(dev)], dim=0)  # noqa: E731
            pc, valid, coors = cat(pa["pcl"], pb["pcl"]), cat(pa["pcl_is_valid"], pb["pcl_is_valid"]), cat(pa["pillar_coors"], pb["pillar_coors"])
            odom = cat(sample_data_t0["gt"]["odom_ta_tb"], sample_data_t1["gt"]["odom_ta_tb"])
            inv_odom = cat(sample_data_t1["gt"]["odom_ta_tb"], sample_data_t0["gt"]["odom_ta_tb"])
            filled = torch.cat([filled0, filled1], dim=0)
            # ... and all RAFT iterations at once: the decoder output of iteration i feeds nothing but the loss, so the
            # 6 x 2B network outputs are decoded as one batch ordered [fw it0..it5 | bw it0..it5]
            n_it = len(out_fw)
            if "net_all" in aux:  # already assembled in that order by one launch (raft_outputs.py)
                net_all = aux["net_all"]
            else:
                batched = aux.get("fw_bw_batched") or [torch.cat([o01, o10], dim=0) for o01, o10 in zip(out_fw, out_bw)]
                net_all = torch.cat([p[:B] for p in batched] + [p[B:] for p in batched], dim=0)
            tile = lambda t: torch.cat([t[:B]] * n_it + [t[B:]] * n_it, dim=0)  # noqa: E731
            pc_all, valid_all, coors_all = tile(pc), tile(valid), tile(coors)
            self.gather_plan_meta = (n_it, tuple(int(v) for v in out_fw[0].shape[1:3]))
            plan = gather_plan if gather_plan is not None else BevGatherPlan.tiled(torch.div(coors, fs, rounding_mode="trunc"), valid,
                                                                                   out_fw[0].shape[1:3], n_it, B)
            out_all = self.head_decoder_fw(net_all, pointwise_valid_mask=valid_all, pointwise_voxel_coordinates=coors_all, pc=pc_all,
                                           filled_pillar_mask=tile(filled), odom=tile(odom), inv_odom=tile(inv_odom),
                                           gather_plan=plan, pointwise_only=self.training and getattr(self, "pointwise_decoding", True),
                                           **common)
            for i in range(n_it):
                preds_fw.append(_slice_prediction(out_all, slice(i * B, (i + 1) * B)))
                preds_bw.append(_slice_prediction(out_all, slice((n_it + i) * B, (n_it + i + 1) * B)))
            # the same predictions stacked over the iterations, for a single evaluation of the per-iteration loss
            self.stacked_predictions = (_slice_prediction(out_all, slice(0, n_it * B)),
                                        _slice_prediction(out_all, slice(n_it * B, 2 * n_it * B)), n_it)
        else:
            plans = [BevGatherPlan(torch.div(sd["pcl_ta"]["pillar_coors"].to(dev), fs, rounding_mode="trunc"),
                                   sd["pcl_ta"]["pcl_is_valid"].to(dev), out_fw[0].shape[1:3])
                     for sd in (sample_data_t0, sample_data_t1)]  # point -> cell lists, shared by all RAFT iterations
            for o01, o10 in zip(out_fw, out_bw):
                kw0 = dict(pointwise_valid_mask=pa["pcl_is_valid"].to(dev), pointwise_voxel_coordinates=pa["pillar_coors"].to(dev),
                           pc=pa["pcl"].to(dev), filled_pillar_mask=filled0, odom=sample_data_t0["gt"]["odom_ta_tb"].to(dev),
                           inv_odom=sample_data_t1["gt"]["odom_ta_tb"].to(dev), gather_plan=plans[0], **common)
                kw1 = dict(pointwise_valid_mask=pb["pcl_is_valid"].to(dev), pointwise_voxel_coordinates=pb["pillar_coors"].to(dev),
                           pc=pb["pcl"].to(dev), filled_pillar_mask=filled1, odom=sample_data_t1["gt"]["odom_ta_tb"].to(dev),
                           inv_odom=sample_data_t0["gt"]["odom_ta_tb"].to(dev), gather_plan=plans[1], **common)
                preds_fw.append(self.head_decoder_fw(o01, **kw0))
                preds_bw.append(self.head_decoder_bw(o10, **kw1))
        self.predictions_fw, self.predictions_bw = preds_fw, preds_bw
        return preds_fw, preds_bw
