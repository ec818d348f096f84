"""RAFT update operator of SLIM.  Mirror of liso/slim/model/update.py:6-164 (same classes, constructor arguments and
attribute names -> same state_dict keys)."""
import os

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

    # ---- inference without concatenations ------------------------------------------------------------------------------------------
    def inference_state(self, net, inp):
        """Buffers of `forward_inference` for one RAFT loop (no autograd): ONE channels-last buffer holds
        [h 96 | inp 64 | out 80 | class 32 | flow 32 | r*h 96] -- channels [0, 304) are the ConvGRU's `hx`, [96, 400) its `[x, r*h]`
        (convq runs on filters whose input channels are permuted accordingly) -- and a second one the motion encoder's
        [corr 96 | class 32 | flow 32].  The convolutions write their channel ranges (mfma_conv.conv2d(out=...)): none of the four
        concatenations of reference :29-37,84-96,139-141 per iteration, and the gates update h in place.  None when not applicable."""
        from liso_amd.utils import mfma_conv as MC

        if (torch.is_grad_enabled() or not net.is_cuda or net.dtype != torch.float32 or not self.predict_logits
                or self.cfg.model.predict_weight_for_static_aggregation or os.environ.get("LISO_UPDATE_SLICES", "1") == "0"):
            return None
        me, gru = self.motion_encoder, self.gru
        ch, ci = net.shape[1], inp.shape[1]
        co, cc, cf = me.conv.out_channels, me.conv_class2.out_channels, me.conv_flow2.out_channels
        cq = me.conv_stat_corr1.out_channels
        if gru.convz.in_channels != ch + ci + co + cc + cf or me.conv.in_channels != cq + cf + cc or any(c % 8 for c in (ch, ci, co, cc, cf, cq)):
            return None
        B, _, H, W = net.shape
        # every convolution that forward_inference runs with `out=` (or on a channel slice) must be one the own kernels cover: the
        # library path cannot write channel ranges (`assert out is None` in mfma_conv.conv2d) -- e.g. correlation planes
        # num_levels * (2 r + 1)^2 that are not a multiple of 4
        probe = lambda conv, c_in: MC.supported(torch.empty((1, c_in, 1, 1), dtype=torch.float32, device=net.device), conv.weight,  # noqa: E731
                                                MC.ConvSpec.of(conv))
        if not (probe(me.conv_stat_corr1, me.conv_stat_corr1.in_channels) and probe(me.conv_flow2, me.conv_flow2.in_channels)
                and probe(me.conv_class2, me.conv_class2.in_channels) and probe(me.conv, cq + cc + cf)
                and probe(gru.convz, gru.convz.in_channels) and probe(gru.convr, gru.convr.in_channels)
                and probe(gru.convq, gru.convq.in_channels)):
            return None
        big = torch.empty((B, H, W, 2 * ch + ci + co + cc + cf), dtype=torch.float32, device=net.device)
        m = torch.empty((B, H, W, cq + cc + cf), dtype=torch.float32, device=net.device)
        big[..., :ch].copy_(net.permute(0, 2, 3, 1))
        big[..., ch:ch + ci].copy_(inp.permute(0, 2, 3, 1))
        x0, x1 = ch, ch + ci + co + cc + cf  # channels of x = (inp, out, class, flow)
        fh, hd = self.static_flow_head, self.classification_head
        tracked = (me.conv.weight, gru.convq.weight, me.conv_flow2.weight, me.conv_class2.weight, me.conv_flow2.bias, me.conv_class2.bias,
                   fh.conv2.weight, hd.conv2.weight, fh.conv2.bias, hd.conv2.bias, me.conv_flow1.weight, me.conv_class1.weight,
                   me.conv_flow1.bias, me.conv_class1.bias)
        key = tuple((t._version, t.data_ptr()) for t in tracked)
        hit = getattr(self, "_infer_perm", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                wc, wq = me.conv.weight, gru.convq.weight
                # `conv` reads (corr, class, flow) instead of the reference's (corr, flow, class); `convq` reads (x, r*h) instead of (r*h, x)
                w_conv = torch.nn.Parameter(torch.cat([wc[:, :cq], wc[:, cq + cf:cq + cf + cc], wc[:, cq:cq + cf]], dim=1).contiguous(),
                                            requires_grad=False)
                w_q = torch.nn.Parameter(torch.cat([wq[:, ch:], wq[:, :ch]], dim=1).contiguous(), requires_grad=False)
                # block-diagonal pairs: ONE launch for conv_class2 | conv_flow2 on the (class1 | flow1) map, ONE for the two heads' output
                # convolutions on the merged hidden map.  The added products are exact zeros and every filter still sees its own
                # channels in their own order: results are bit for bit those of the separate convolutions.
                c1c, c1f = me.conv_class2.in_channels, me.conv_flow2.in_channels
                w_cf = torch.zeros((cc + cf, c1c + c1f, 3, 3), dtype=wc.dtype, device=wc.device)
                w_cf[:cc, :c1c] = me.conv_class2.weight
                w_cf[cc:, c1c:] = me.conv_flow2.weight
                b_cf = torch.cat([me.conv_class2.bias, me.conv_flow2.bias])
                of, oc, hf, hc = fh.conv2.out_channels, hd.conv2.out_channels, fh.conv2.in_channels, hd.conv2.in_channels
                w_hd = torch.zeros((of + oc, hf + hc, 3, 3), dtype=wc.dtype, device=wc.device)
                w_hd[:of, :hf] = fh.conv2.weight
                w_hd[of:, hf:] = hd.conv2.weight
                b_hd = torch.cat([fh.conv2.bias, hd.conv2.bias])
                # conv_class1 | conv_flow1 (7x7 on 4 / 2 channels, :54-59) as ONE launch on the loop's packed state pixel
                # (flow x, flow y, logit 0..3, 0, 0): the kernels pad 2-4 input channels to a 16-channel slab anyway
                w_71 = b_71 = None
                if (tuple(me.conv_class1.weight.shape[1:]) == (4, 7, 7) and tuple(me.conv_flow1.weight.shape[1:]) == (2, 7, 7)
                        and me.conv_class1.padding == me.conv_flow1.padding and of == 2 and oc == 4):
                    k1c, k1f = me.conv_class1.out_channels, me.conv_flow1.out_channels
                    w_71 = torch.zeros((k1c + k1f, 8, 7, 7), dtype=wc.dtype, device=wc.device)
                    w_71[:k1c, 2:6] = me.conv_class1.weight
                    w_71[k1c:, 0:2] = me.conv_flow1.weight
                    b_71 = torch.cat([me.conv_class1.bias, me.conv_flow1.bias])
                P = lambda t: None if t is None else torch.nn.Parameter(t.contiguous(), requires_grad=False)  # noqa: E731
            hit = self._infer_perm = (key, w_conv, w_q, P(w_cf), P(b_cf), P(w_hd), P(b_hd), P(w_71), P(b_71))
        merged = (os.environ.get("LISO_UPDATE_MERGED", "1") != "0" and tuple(me.conv_class2.kernel_size) == (3, 3)
                  and tuple(me.conv_flow2.kernel_size) == (3, 3) and tuple(fh.conv2.kernel_size) == (3, 3) and tuple(hd.conv2.kernel_size) == (3, 3)
                  and me.conv_class1.out_channels % 8 == 0 and me.conv_flow1.out_channels % 8 == 0)
        # the loop's (flow | logits) state as one channels-last pixel of 8 floats, updated by ONE launch per iteration (state_step)
        packed = merged and hit[7] is not None and os.environ.get("LISO_UPDATE_PACKED_STATE", "1") != "0"
        return {"big": big, "m": m, "ch": ch, "ci": ci, "co": co, "cc": cc, "cf": cf, "cq": cq, "x0": x0, "x1": x1,
                "w_conv": hit[1], "w_q": hit[2], "w_cf": hit[3], "b_cf": hit[4], "w_hd": hit[5], "b_hd": hit[6], "merged": merged,
                "w_71": hit[7], "b_71": hit[8], "packed": packed,
                "fl8": torch.zeros((B, H, W, 8), dtype=torch.float32, device=net.device) if packed else None,
                "c1": torch.empty((B, H, W, me.conv_class1.out_channels + me.conv_flow1.out_channels), dtype=torch.float32, device=net.device)
                if merged else None,
                "z": torch.empty((B, H, W, ch), dtype=torch.float32, device=net.device)}

    def state_step(self, st, both, coords0, coords1):
        """(packed state) reference raft.py:199-216 in one launch: coords1 += delta_flow (in place), logits += delta_logits and
        flow = coords1 - coords0 into st["fl8"]; `both` = the heads' merged output [B, 2 + 4, h, w].  -> logits view [B,4,h,w]"""
        import ctypes  # noqa: F401

        from liso_amd import _lib as L
        from liso_amd.utils import mfma_conv as MC

        fl8 = st["fl8"]
        bv, bps = MC.as_nhwc(both, 2)
        B, H, W, _ = fl8.shape
        assert coords0.is_contiguous() and coords1.is_contiguous() and coords1.shape == (B, 2, H, W) and both.shape[1] == 6
        with torch.cuda.device(fl8.device):
            L.check(L.lib().liso_raft_state_step_f32(B, H * W, L.ptr(bv), bps, L.ptr(coords0), L.ptr(coords1), L.ptr(fl8), L.stream_ptr()),
                    "raft_state_step")
        return fl8[..., 2:6].permute(0, 3, 1, 2)

    def forward_inference(self, st, corr, flow, logits):
        """one update iteration on the buffers of `inference_state` -> (net view, delta_static_flow, delta_logits, None); with the
        packed state (st["packed"]: `flow` / `logits` are read from st["fl8"]) -> (net view, merged head output [B, 2 + 4, h, w])"""
        import ctypes
        import types

        from liso_amd import _lib as L
        from liso_amd.utils import mfma_conv as MC

        me, gru = self.motion_encoder, self.gru
        big, m = st["big"], st["m"]
        ch, ci, co, cc, cf, cq = st["ch"], st["ci"], st["co"], st["cc"], st["cf"], st["cq"]
        o_out, o_cls, o_flow, o_rh = ch + ci, ch + ci + co, ch + ci + co + cc, st["x1"]
        bign, mn = big.permute(0, 3, 1, 2), m.permute(0, 3, 1, 2)
        MC.conv2d(me.conv_stat_corr1, corr, relu=True, out=(m, 0))
        if st["merged"]:  # (class1 | flow1) into one map, then conv_class2 | conv_flow2 as ONE block-diagonal launch -> big[class | flow]
            c1 = st["c1"]
            k1 = me.conv_class1.out_channels
            if st["packed"]:
                MC.fused_conv(st["fl8"].permute(0, 3, 1, 2), None, types.SimpleNamespace(weight=st["w_71"], bias=st["b_71"]), out_relu=True,
                              spec=MC.ConvSpec.of(me.conv_class1), out=(c1, 0))
            else:
                MC.conv2d(me.conv_class1, logits, relu=True, out=(c1, 0))
                MC.conv2d(me.conv_flow1, flow, relu=True, out=(c1, k1))
            MC.fused_conv(c1.permute(0, 3, 1, 2), None, types.SimpleNamespace(weight=st["w_cf"], bias=st["b_cf"]), out_relu=True,
                          spec=MC.ConvSpec.of(me.conv_class2), out=(big, o_cls))
        else:
            MC.conv2d(me.conv_flow2, MC.conv2d(me.conv_flow1, flow, relu=True), relu=True, out=(big, o_flow))
            MC.conv2d(me.conv_class2, MC.conv2d(me.conv_class1, logits, relu=True), relu=True, out=(big, o_cls))
        m[..., cq:].copy_(big[..., o_cls:o_rh])  # (class, flow): the one copy left, 64 of the 656 channels the concatenations moved
        MC.fused_conv(mn, None, types.SimpleNamespace(weight=st["w_conv"], bias=me.conv.bias), out_relu=True, spec=MC.ConvSpec.of(me.conv),
                      out=(big, o_out))
        zr, _ = MC.fused_conv(bign[:, :o_rh], None, [gru.convz, gru.convr])
        zrv, zps = MC.as_nhwc(zr, 4)
        n_pix, wt = big.shape[0] * big.shape[1] * big.shape[2], big.shape[3]
        z = st["z"]
        lib = L.lib()
        with torch.cuda.device(big.device):
            L.check(lib.liso_gru_in_rows_f32(n_pix, ch, L.ptr(zrv), zps, L.ptr(big), wt, L.ptr(z), ctypes.c_void_p(big.data_ptr() + 4 * o_rh),
                                             wt, L.stream_ptr()), "gru_in_rows")
        cqv, _ = MC.fused_conv(bign[:, ch:], None, types.SimpleNamespace(weight=st["w_q"], bias=gru.convq.bias), spec=MC.ConvSpec.of(gru.convq))
        cqn, cps = MC.as_nhwc(cqv, 4)
        with torch.cuda.device(big.device):
            L.check(lib.liso_gru_out_rows_f32(n_pix, ch, L.ptr(cqn), cps, L.ptr(z), L.ptr(big), wt, L.stream_ptr()), "gru_out_rows")
        net = bign[:, :ch]
        fh, hd = self.static_flow_head, self.classification_head
        hid = conv2d_pair(fh.conv1, hd.conv1, net, relu=True)
        if st["merged"]:  # both heads' output convolutions as ONE block-diagonal launch on the merged hidden map
            both, _ = MC.fused_conv(hid, None, types.SimpleNamespace(weight=st["w_hd"], bias=st["b_hd"]), spec=MC.ConvSpec.of(fh.conv2))
            if st["packed"]:
                return net, both
            of = fh.conv2.out_channels
            return net, both[:, :of], both[:, of:], None
        hid_f, hid_c = torch.split(hid, [fh.conv1.out_channels, hd.conv1.out_channels], dim=1)
        return net, conv2d(fh.conv2, hid_f), conv2d(hd.conv2, hid_c), None

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
