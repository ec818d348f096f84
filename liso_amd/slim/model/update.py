"""RAFT update operator of SLIM.  Mirror of liso/slim/model/update.py:6-164 (same classes, constructor arguments and
attribute names -> same state_dict keys)."""
import torch
import torch.nn.functional as F
from torch import nn

from liso_amd.slim.model.deferred_wgrad import conv2d, conv2d_pair
from liso_amd.slim.model.gru_gates import gru_in, gru_out


class FlowOrClassificationHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, out_dims=2, **kwargs):
        super().__init__(**kwargs)
        assert out_dims in [2, 3, 4]
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, out_dims, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, inputs):
        return conv2d(self.conv2, conv2d(self.conv1, inputs, relu=True))


class ConvGRU(nn.Module):
    def __init__(self, hidden_dim=96, input_dim=304):
        super().__init__()
        self.convz = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.convr = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.convq = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.merged_convs = [(self.convz, self.convr)]  # same input: one convolution with 2 x hidden_dim outputs
        self.fused_gates = True

    def forward(self, h, x):
        """reference :29-37.  On the GPU (fp32): convz and convr as one convolution, the gate arithmetic in two fused
        launches (gru_gates.py); otherwise the reference's op sequence."""
        hx = torch.cat([h, x], dim=1)
        if self.fused_gates and h.is_cuda and h.dtype == torch.float32:
            z, rhx = gru_in(conv2d_pair(self.convz, self.convr, hx), h, x)
            return gru_out(conv2d(self.convq, rhx), z, h)
        z = torch.sigmoid(conv2d(self.convz, hx))
        r = torch.sigmoid(conv2d(self.convr, hx))
        q = torch.tanh(conv2d(self.convq, torch.cat([r * h, x], dim=1)))
        return (1 - z) * h + z * q


class SmallMotionEncoder(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__(**kwargs)
        self.cfg = cfg
        corr_planes = cfg.model.corr_cfg.num_levels * (2 * cfg.model.corr_cfg.search_radius + 1) ** 2
        self.conv_stat_corr1 = nn.Conv2d(corr_planes, 96, 1, padding=0)
        flow_ch = 3 if cfg.model.predict_weight_for_static_aggregation is not False else 2
        self.conv_flow1 = nn.Conv2d(flow_ch, 64, 7, padding=3)
        self.conv_flow2 = nn.Conv2d(64, 32, 3, padding=1)
        self.predict_logits = cfg.model.flow_maps_archi != "vanilla"
        if self.predict_logits:
            self.conv_class1 = nn.Conv2d(4, 64, 7, padding=3)
            self.conv_class2 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv = nn.Conv2d(128 + int(self.predict_logits) * 32, 80, 3, padding=1)

    def forward(self, flow, corr, logits):
        """reference :74-96"""
        corr = conv2d(self.conv_stat_corr1, corr, relu=True)
        flow = conv2d(self.conv_flow2, conv2d(self.conv_flow1, flow, relu=True), relu=True)
        vals = [corr, flow]
        if self.predict_logits:
            logits = conv2d(self.conv_class2, conv2d(self.conv_class1, logits, relu=True), relu=True)
            vals.append(logits)
        else:
            assert logits is None
        out = conv2d(self.conv, torch.cat(vals, dim=1), relu=True)
        if self.predict_logits:
            return torch.cat([out, logits, flow], dim=1)
        return torch.cat([out, flow], dim=1)


class SmallUpdateBlock(nn.Module):
    def __init__(self, cfg, filters=96, **kwargs):
        super().__init__(**kwargs)
        self.cfg = cfg
        self.filters = filters
        self.predict_logits = cfg.model.flow_maps_archi != "vanilla"
        self.motion_encoder = SmallMotionEncoder(cfg=cfg)
        self.gru = ConvGRU(hidden_dim=filters, input_dim=272 + int(self.predict_logits) * 32)
        flow_ch = 3 if cfg.model.predict_weight_for_static_aggregation is not False else 2
        self.static_flow_head = FlowOrClassificationHead(input_dim=filters, hidden_dim=128, out_dims=flow_ch)
        self.classification_head = (FlowOrClassificationHead(input_dim=filters, hidden_dim=128, out_dims=4)
                                    if self.predict_logits else None)
        # both heads read the hidden state through a 3x3 convolution of the same geometry: ONE convolution with 2 x 128 filters
        # (one staging of `net`, one data gradient instead of two and their sum); every head's output convolution then reads its
        # 128-channel slice of that map
        self.merged_convs = [(self.static_flow_head.conv1, self.classification_head.conv1)] if self.predict_logits else []
        self.merge_head_convs = True

    def forward(self, net, inp, corr, flow, logits, weight_logits_for_static_aggregation):
        """reference :130-164"""
        if self.cfg.model.predict_weight_for_static_aggregation:
            mf = self.motion_encoder(torch.cat([flow, weight_logits_for_static_aggregation], dim=1), corr, logits)
        else:
            assert weight_logits_for_static_aggregation is None
            mf = self.motion_encoder(flow, corr, logits)
        net = self.gru(net, torch.cat([inp, mf], dim=1))
        if self.merged_convs and self.merge_head_convs and net.is_cuda:
            fh, ch = self.static_flow_head, self.classification_head
            hid = conv2d_pair(fh.conv1, ch.conv1, net, relu=True)
            # (split, not two slices: its backward is one concatenation of the heads' input gradients)
            hid_f, hid_c = torch.split(hid, [fh.conv1.out_channels, ch.conv1.out_channels], dim=1)
            delta, delta_logits = conv2d(fh.conv2, hid_f), conv2d(ch.conv2, hid_c)
        else:
            delta = self.static_flow_head(net)
            delta_logits = self.classification_head(net) if self.predict_logits else None
        if self.cfg.model.predict_weight_for_static_aggregation:
            delta_static_flow, delta_weights = delta[:, 0:2, ...], delta[:, -1:, ...]
        else:
            delta_static_flow, delta_weights = delta, None
        return net, delta_static_flow, delta_logits, delta_weights
