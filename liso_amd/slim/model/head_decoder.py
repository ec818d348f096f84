"""HeadDecoder of SLIM.  Mirror of liso/slim/model/head_decoder.py (round 1: the network-output packing
`concat2network_output`, :37-65; the per-point decoding of :67-496 lands with the SLIM loss rows)."""
import torch
from torch import nn


class HeadDecoder(nn.Module):
    def __init__(self, cfg, name, bev_extent, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.cfg, self.name, self.bev_extent = cfg, name, bev_extent

    def concat2network_output(self, *, logits, static_flow, dynamic_flow, weight_logits_for_static_aggregation=None):
        assert logits.shape[1] == 4 and static_flow.shape[1] == 2 and dynamic_flow.shape[1] == 2
        assert (weight_logits_for_static_aggregation is None) == (not self.cfg.model.predict_weight_for_static_aggregation)
        parts = [logits, static_flow, dynamic_flow]
        if weight_logits_for_static_aggregation is not None:
            assert weight_logits_for_static_aggregation.shape[1] == 1
            parts.append(weight_logits_for_static_aggregation)
        return torch.cat(parts, dim=1).permute(0, 2, 3, 1)
