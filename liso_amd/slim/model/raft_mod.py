"""RAFT-on-BEV of SLIM.  Mirror of liso/slim/model/raft_mod.py:19-266 (same constructor, sub-module names `pp_layer`,
`fnet`, `cnet`, `update_block`, same forward signature and outputs).  The pillar encoder is the fused gfx950 pillar
path, the correlation lookup the on-the-fly gfx950 kernel; convolutions run channels-last through MIOpen."""
import numpy as np
import torch
from torch import nn

from liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid import PointsPillarFeatureNetWrapper
from liso_amd.slim.model.extractor import SmallEncoder
from liso_amd.slim.model.raft_code.corr import CorrBlock
from liso_amd.slim.model.deferred_wgrad import deferred_weight_gradients
from liso_amd.slim.model.raft_outputs import raft_network_outputs
from liso_amd.slim.model.raft_code.utils import initialize_flow, upflow_n, uplogits_n
from liso_amd.slim.model.update import SmallUpdateBlock


def move_channel_to_last_dim(tensor):
    return tensor.permute(0, 2, 3, 1)


def change_flow_convention_from_raft2usfl(flow, resolution_adapter):
    """reference :262-266 -- RAFT's (x=col, y=row) pixel flow -> (row, col) metres"""
    return torch.flip(flow, dims=[1]) * resolution_adapter


def _in_passes(net, x, occ, images_per_pass, half=False):
    """encoder `net` on the batch `x` in passes of at most `images_per_pass` images (None: one pass).  Inference batches of many
    sweep pairs: the update block gains from every extra pair per launch, the encoders' 256^2 x 32-channel fp32 activations
    (8.4 MB per image and tensor) stop fitting the 256-MB Infinity Cache beyond ~8 images per pass.  Per-sample normalisation
    (InstanceNorm) or none: the result does not depend on the split."""
    n = images_per_pass
    if n is not None and half:
        n = max(1, n // 2)
    if n is None or x.shape[0] <= n:
        return net(x, occupancy=occ) if occ is not None else net(x)
    outs = [net(x[k:k + n], occupancy=occ[k:k + n]) if occ is not None else net(x[k:k + n]) for k in range(0, x.shape[0], n)]
    return torch.cat(outs, dim=0)


class RAFT(nn.Module):
    def __init__(self, cfg, head_decoder_fw, head_decoder_bw, **kwargs):
        super().__init__(**kwargs)
        self.cfg = cfg
        self.slim_cfg = cfg.SLIM
        half = 0.5 * np.array(cfg.data.bev_range_m)
        bev_pc_range = np.concatenate([-half, half], axis=0)
        self.head_decoder_fw, self.head_decoder_bw = head_decoder_fw, head_decoder_bw
        self.iters = self.slim_cfg.model.num_iters
        fs = self.slim_cfg.model.u_net.final_scale
        self.bev_rows_res_meters_per_fs_pixel = (bev_pc_range[2] - bev_pc_range[0]) / cfg.data.img_grid_size[0] * fs
        self.bev_cols_res_meters_per_fs_pixel = (bev_pc_range[3] - bev_pc_range[1]) / cfg.data.img_grid_size[1] * fs
        assert self.bev_rows_res_meters_per_fs_pixel == self.bev_cols_res_meters_per_fs_pixel
        self.pp_layer = PointsPillarFeatureNetWrapper(cfg)
        self.hidden_dim, self.context_dim = 96, 64
        assert self.slim_cfg.model.corr_cfg.module == "all" and self.slim_cfg.model.feature_downsampling_factor == 8
        self.fnet = SmallEncoder(output_dim=128, norm_fn=self.slim_cfg.model.raft_fnet_norm,
                                 dropout=self.slim_cfg.model.dropout_rate)
        self.cnet = SmallEncoder(output_dim=self.hidden_dim + self.context_dim, norm_fn="none",
                                 dropout=self.slim_cfg.model.dropout_rate)
        self.update_block = SmallUpdateBlock(cfg=self.slim_cfg, filters=self.hidden_dim)

    def forward(self, pcl_t0, pcl_t1, canvases=None):
        """reference :82-122.  The forward (t0->t1) and backward (t1->t0) flow estimates share every weight and never
        interact inside the network (instance norm in `fnet`, no norm elsewhere), so they run as ONE batch of 2B samples
        through the encoders, the correlation lookup and the 6 update iterations: per-sample results are those of the
        reference's two sequential calls, with half the launches and twice the work per convolution (the 64x64 update
        maps of a single sample cannot fill 256 CUs).  The pillar encoder stays per sweep: its BatchNorm1d statistics are
        per call in the reference."""
        if canvases is not None:  # (extension) precomputed pillar canvases: callers that replay the rest from a hipGraph
            img_t0, occ_t0, img_t1, occ_t1 = canvases
        else:
            img_t0, occ_t0 = self.pp_layer(pcl_t0)
            img_t1, occ_t1 = self.pp_layer(pcl_t1)
        aux = {"t0": {"bev_net_input_dbg": occ_t0}, "t1": {"bev_net_input_dbg": occ_t1}}
        if not getattr(self, "batch_directions", True):  # the reference's schedule: two sequential passes (:95-121)
            o0, o1 = (occ_t0, occ_t1) if img_t0.is_cuda else (None, None)  # (the same sparse-canvas stem as the batched schedule)
            fmap_t0, fmap_t1 = self.fnet(img_t0, occupancy=o0), self.fnet(img_t1, occupancy=o1)
            fw = self.predict_single_flow_map_and_classes(img_t0, fmap_t0, fmap_t1, self.head_decoder_fw, occupancy_t0=o0)
            bw = self.predict_single_flow_map_and_classes(img_t1, fmap_t1, fmap_t0, self.head_decoder_bw, occupancy_t0=o1)
            return fw, bw, aux
        B = img_t0.shape[0]
        imgs = torch.cat([img_t0, img_t1], dim=0)
        occ = torch.cat([occ_t0, occ_t1], dim=0) if imgs.is_cuda else None  # (the sparse-canvas paths of the encoders' first convolution)
        fmap = self.fnet(imgs, occupancy=occ)
        fmap_swapped = torch.cat([fmap[B:], fmap[:B]], dim=0)
        both = self.predict_single_flow_map_and_classes(imgs, fmap, fmap_swapped, self.head_decoder_fw,
                                                        fused_dirs=2 if getattr(self, "fused_outputs", True) else None,
                                                        occupancy_t0=occ)
        if torch.is_tensor(both):  # all iterations assembled by one launch: [fw it0..itN | bw it0..itN] x B samples
            n_it = both.shape[0] // (2 * B)
            aux["net_all"] = both
            return ([both[i * B:(i + 1) * B] for i in range(n_it)],
                    [both[(n_it + i) * B:(n_it + i + 1) * B] for i in range(n_it)], aux)
        aux["fw_bw_batched"] = both  # per iteration [2B,H,W,8]: samples [:B] = forward flow, [B:] = backward flow
        return [p[:B] for p in both], [p[B:] for p in both], aux

    def encode_pillars(self, pcl_t0, pcl_t1, out=None):
        """the pillar canvases of both sweeps: (img_t0, occ_t0, img_t1, occ_t1); differentiable w.r.t. the pillar encoder's
        parameters when gradients are enabled.  `out` = (rows [2B, gx, gy, 64], occupancy [2B, 1, gx, gy]): both sweeps are
        written into these buffers (static hipGraph inputs: no copy, and the batch of both sweeps needs no concatenation);
        the result then carries a fifth element, the stacked canvas [2B, 64, gx, gy]."""
        if out is None:
            return (*self.pp_layer(pcl_t0), *self.pp_layer(pcl_t1))
        rows, occ = out
        B = rows.shape[0] // 2
        a = self.pp_layer(pcl_t0, out=(rows[:B], occ[:B]))
        b = self.pp_layer(pcl_t1, out=(rows[B:], occ[B:]))
        return (*a, *b, rows.permute(0, 3, 1, 2))

    def infer_forward_direction(self, pcl_t0, pcl_t1, canvases=None):
        """Inference for consumers of the t0 -> t1 flow only (the box miner): one direction, last iteration.
        -> ([B,H,W,8(+1)] network output, aux).  `canvases`: precomputed `encode_pillars` result (callers that replay the
        rest from a hipGraph keep the pillar encoder outside of it)."""
        canvases = canvases if canvases is not None else self.encode_pillars(pcl_t0, pcl_t1)
        img_t0, occ_t0, img_t1, occ_t1 = canvases[:4]
        aux = {"t0": {"bev_net_input_dbg": occ_t0}, "t1": {"bev_net_input_dbg": occ_t1}}
        B = img_t0.shape[0]
        # (canvases[4] / [5]: both sweeps stacked along the batch axis and their occupancy maps, when the caller holds them that way)
        occ_all = canvases[5] if len(canvases) > 5 else (torch.cat([occ_t0, occ_t1], dim=0) if img_t0.is_cuda else None)
        fmap = _in_passes(self.fnet, canvases[4] if len(canvases) > 4 else torch.cat([img_t0, img_t1], dim=0), occ_all,
                          getattr(self, "encoder_images_per_pass", None))
        out = self.predict_single_flow_map_and_classes(img_t0, fmap[:B], fmap[B:], self.head_decoder_fw, only_last=True,
                                                       occupancy_t0=occ_t0)
        return out[-1], aux

    def infer_both_directions(self, pcl_t0, pcl_t1, canvases=None):
        """Inference for the flow EXPORT (liso/slim/experiment.py:363-471 writes `bev_raw_flow_t0_t1` AND `bev_raw_flow_t1_t0`): both
        flow directions as one batch of 2B samples like forward(), last RAFT iteration only.
        -> ([B,H,W,8] forward output, [B,H,W,8] backward output, aux)"""
        canvases = canvases if canvases is not None else self.encode_pillars(pcl_t0, pcl_t1)
        img_t0, occ_t0, img_t1, occ_t1 = canvases[:4]
        aux = {"t0": {"bev_net_input_dbg": occ_t0}, "t1": {"bev_net_input_dbg": occ_t1}}
        B = img_t0.shape[0]
        imgs = canvases[4] if len(canvases) > 4 else torch.cat([img_t0, img_t1], dim=0)
        occ = (canvases[5] if len(canvases) > 5 else torch.cat([occ_t0, occ_t1], dim=0)) if imgs.is_cuda else None
        fmap = self.fnet(imgs, occupancy=occ)
        out = self.predict_single_flow_map_and_classes(imgs, fmap, torch.cat([fmap[B:], fmap[:B]], dim=0), self.head_decoder_fw,
                                                       only_last=True, occupancy_t0=occ)
        return out[-1][:B], out[-1][B:], aux

    def predict_single_flow_map_and_classes(self, img_t0, fmap_t0, fmap_t1, decoder, only_last=False, fused_dirs=None,
                                            occupancy_t0=None):
        """reference :124-259.  `only_last` (extension, inference): upsample / assemble the network output of the last
        iteration only -- the intermediate ones exist for the training loss.  `fused_dirs` (extension): the batch holds
        `fused_dirs` flow directions x B samples; the outputs of all iterations are then assembled by one launch
        (raft_outputs.py) and returned as ONE tensor [fused_dirs * n_it * B, H, W, 8] instead of a list."""
        m = self.slim_cfg.model
        assert img_t0.shape[1] == m.point_pillars.nbr_point_feats, img_t0.shape
        ds = m.feature_downsampling_factor
        coords0 = initialize_flow(img_t0, downscale_factor=ds)
        coords1 = initialize_flow(img_t0, downscale_factor=ds)
        b, _, h, w = coords0.shape
        vanilla = m.flow_maps_archi == "vanilla"
        logits = None if vanilla else torch.zeros((b, 4, h, w), dtype=torch.float32, device=img_t0.device)
        use_w = m.predict_weight_for_static_aggregation is not False
        wl = torch.zeros((b, 1, h, w), dtype=torch.float32, device=img_t0.device) if use_w else None
        correlation = CorrBlock(fmap_t0, fmap_t1, num_levels=m.corr_cfg.num_levels, radius=m.corr_cfg.search_radius)
        if occupancy_t0 is not None:
            cnet = _in_passes(self.cnet, img_t0, occupancy_t0, getattr(self, "encoder_images_per_pass", None) if only_last else None, half=True)
        else:
            cnet = self.cnet(img_t0)
        net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        # (rows, cols) metres per pixel; equal (asserted in __init__), so a python scalar does the job of the reference's
        # [1,2,1,1] tensor (:171-176) without a host->device copy per call
        adapter = float(self.bev_rows_res_meters_per_fs_pixel)
        preds = []
        fuse = (fused_dirs is not None and not vanilla and not use_w and not only_last and img_t0.is_cuda
                and b % fused_dirs == 0)
        lowres_flows, lowres_logits = [], []
        # weight gradients of the update block: one convolution per layer over all iterations (deferred_wgrad.py)
        mode = getattr(self, "defer_update_block_wgrad", True)  # True | False | "direct" (hipGraph capture, see trainer.py)
        infer = self.update_block.inference_state(net, inp) if (only_last and not use_w) else None  # (inference: no concatenations)
        packed = infer is not None and infer["packed"] and not vanilla  # (flow | logits) in one state pixel, one update launch per iteration
        if packed:
            coords0, coords1 = coords0.contiguous(), coords1.contiguous()  # (coords_grid returns fresh tensors: coords1 is updated in place)
        if fuse and infer is None and torch.is_grad_enabled():
            # training: ALL iterations as one autograd node on stacked buffers, hand-sequenced backward (raft_loop.py); the op-by-op
            # loop below remains for the configurations the node does not implement (and as its test oracle: LISO_RAFT_LOOP=0)
            from liso_amd.slim.model import raft_loop as RL
            from liso_amd.slim.model.raft_outputs import _RaftOutputs

            if RL.applicable(self.update_block, net, inp, m.corr_cfg.search_radius):
                flows, logits_all = RL.raft_loop(self.update_block, correlation, coords0, net, inp, m.num_iters, direct=mode == "direct")
                return _RaftOutputs.apply(flows, logits_all, fused_dirs, ds, ds * float(adapter))
        with deferred_weight_gradients(self.update_block, enabled=self.training and bool(mode), direct_accumulate=mode == "direct"):
            for it in range(m.num_iters):
                coords1 = coords1.detach()
                if not vanilla:
                    logits = logits.detach()
                if use_w:
                    wl = wl.detach()
                corr = correlation(coords1)
                if packed:
                    net, both = self.update_block.forward_inference(infer, corr, None, None)
                    logits = self.update_block.state_step(infer, both, coords0, coords1)
                    if it + 1 < m.num_iters:
                        continue
                    logits = logits.contiguous()
                else:
                    flow = coords1 - coords0
                    if infer is not None:
                        net, d_flow, d_logits, d_w = self.update_block.forward_inference(infer, corr, flow, logits)
                    else:
                        net, d_flow, d_logits, d_w = self.update_block(net, inp, corr, flow, logits, wl)
                    coords1 = coords1 + d_flow
                    if not vanilla:
                        logits = logits + d_logits
                    if use_w:
                        wl = wl + d_w
                if only_last and it + 1 < m.num_iters:
                    continue
                if fuse:
                    lowres_flows.append(coords1 - coords0)
                    lowres_logits.append(logits)
                    continue
                if only_last and not vanilla and not use_w and img_t0.is_cuda and not torch.is_grad_enabled():
                    # inference: the one output that is read, assembled by the launch the training step uses for all iterations
                    # (x8 upsampling + flow convention + concatenation + channels-last: 7 framework launches on 512^2 maps otherwise)
                    from liso_amd.slim.model.raft_outputs import _RaftOutputs

                    preds.append(_RaftOutputs.apply((coords1 - coords0)[None], logits.contiguous()[None], 1, ds, ds * adapter))
                    continue
                up_flow = change_flow_convention_from_raft2usfl(upflow_n(coords1 - coords0, n=ds), resolution_adapter=adapter)
                if vanilla:
                    up_logits = torch.zeros((b, 4, h * ds, w * ds), dtype=torch.float32, device=img_t0.device)
                else:
                    up_logits = uplogits_n(logits, n=ds)
                up_w = uplogits_n(wl, n=ds) if use_w else None
                preds.append(decoder.concat2network_output(logits=up_logits, static_flow=up_flow, dynamic_flow=up_flow,
                                                           weight_logits_for_static_aggregation=up_w))
        if fuse:
            return raft_network_outputs(lowres_flows, lowres_logits, dirs=fused_dirs, factor=ds, resolution_adapter=adapter)
        return preds
