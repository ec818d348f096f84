"""RPN: SECOND-style BEV backbone.  Mirror of liso/networks/centerpoint/rpn.py (same ctor, same Sequential indices ->
same state_dict keys `blocks.{i}.{j}.*`, `deblocks.{i}.{j}.*`, including the ZeroPad2d at index 0, rpn.py:113-129).

MI355X notes: tensors stay channels-last (the pillar scatter already produces NHWC storage) and in the dtype they
arrive in (bf16 for the perf configuration, fp32 for parity); every convolution -- forward, data gradient, weight
gradient -- runs on the library's own MFMA implicit-GEMM kernels (include/liso_conv.h: conv_roles_kernel for the
3x3 / stride-1 layers, conv_igemm_kernel for the rest, the sparse-canvas kernels for the first layer), with the
BatchNorm-apply + ReLU of the producing layer as the consumer's prologue and the batch statistics as the producer's
epilogue (fp32 statistics; include/liso_bn.h for the backward passes).  forward() is functional so ZeroPad2d +
conv(pad=0) becomes one padded conv (identical arithmetic, one HBM round trip less per stage).  Host tensors (the CPU test tier)
step through liso_amd/utils/host_ops.py; there is no other route for a device tensor than the own kernels.
"""
import numpy as np
import torch
from torch import nn

from liso_amd.networks.centerpoint.fused_bn import bn_act
from liso_amd.networks.centerpoint.norm import baurst_build_norm_layer as build_norm_layer
from liso_amd.networks.centerpoint.weight_init import xavier_init


def conv_bn_relu(x, conv, bn, stride=None, padding=None):
    """conv (+bias) -> BatchNorm2d -> ReLU on a HOST tensor (CPU test tier); BN stats in fp32 whatever the conv dtype."""
    from liso_amd.utils import host_ops

    w = conv.weight
    if w.dtype != x.dtype:
        w = w.to(x.dtype)
    b = conv.bias.to(x.dtype) if conv.bias is not None else None
    if isinstance(conv, nn.ConvTranspose2d):
        y = host_ops.conv_transpose2d(x, w, b, stride=conv.stride)
    else:
        y = host_ops.conv2d(x, w, b, stride=conv.stride if stride is None else stride,
                            padding=conv.padding if padding is None else padding)
    return bn_act(y, bn, relu=True)


class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", **kwargs):
        super().__init__()
        self._layer_strides = ds_layer_strides
        self._num_filters = ds_num_filters
        self._layer_nums = layer_nums
        self._upsample_strides = us_layer_strides
        self._num_upsample_filters = us_num_filters
        self._num_input_features = num_input_features
        if norm_cfg is None:
            norm_cfg = {"type": "BN", "eps": 1e-3, "momentum": 0.01}  # reference :35-36
        self._norm_cfg = norm_cfg
        assert len(self._layer_strides) == len(self._layer_nums) == len(self._num_filters)
        assert len(self._num_upsample_filters) == len(self._upsample_strides)
        self._upsample_start_idx = len(self._layer_nums) - len(self._upsample_strides)
        in_filters = [self._num_input_features, *self._num_filters[:-1]]
        blocks, deblocks = [], []
        for i, layer_num in enumerate(self._layer_nums):
            block, num_out = self._make_layer(in_filters[i], self._num_filters[i], layer_num,
                                              stride=self._layer_strides[i])
            blocks.append(block)
            if i - self._upsample_start_idx >= 0:  # reference :70-104
                stride = self._upsample_strides[i - self._upsample_start_idx]
                out_f = self._num_upsample_filters[i - self._upsample_start_idx]
                if stride > 1:
                    up = nn.ConvTranspose2d(num_out, out_f, stride, stride=stride, bias=False)
                else:
                    k = int(np.round(1 / stride).astype(np.int64))
                    up = nn.Conv2d(num_out, out_f, k, stride=k, bias=False)
                deblocks.append(nn.Sequential(up, build_norm_layer(self._norm_cfg, out_f)[1], nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)

    @property
    def downsample_factor(self):
        factor = np.prod(self._layer_strides)
        if len(self._upsample_strides) > 0:
            factor /= self._upsample_strides[-1]
        return factor

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        """reference :113-131"""
        layers = [nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False),
                  build_norm_layer(self._norm_cfg, planes)[1], nn.ReLU()]
        for _ in range(num_blocks):
            layers += [nn.Conv2d(planes, planes, 3, padding=1, bias=False),
                       build_norm_layer(self._norm_cfg, planes)[1], nn.ReLU()]
        return nn.Sequential(*layers), planes

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution="uniform")

    def forward(self, x, lazy=False, occupancy=None):
        """reference :137-146.  On the GPU every convolution runs on the own MFMA kernels with the BatchNorm+ReLU of layer k
        folded into the prologue of layer k+1 (liso_amd/utils/mfma_conv.py); `lazy=True` (CenterPointStyleNet) returns
        (raw concatenated maps, BnFold) for CenterHead instead of materialising the normalised feature map."""
        from liso_amd.utils import mfma_conv as MC

        if MC.on_device(x):
            raw, fold = self._forward_fused(x, MC, occupancy)
            return (raw, fold) if lazy else MC.materialize(raw, fold)
        ups = []
        for i, block in enumerate(self.blocks):
            mods = list(block)
            x = conv_bn_relu(x, mods[1], mods[2], padding=1)  # ZeroPad2d(1) + conv(pad 0) == conv(pad 1)
            for j in range(4, len(mods), 3):
                x = conv_bn_relu(x, mods[j], mods[j + 1])
            if i - self._upsample_start_idx >= 0:
                d = self.deblocks[i - self._upsample_start_idx]
                ups.append(conv_bn_relu(x, d[0], d[1]))
        if len(ups) > 0:
            x = torch.cat(ups, dim=1)
        return x

    def _forward_fused(self, x, MC, occupancy=None):
        """`occupancy` (extension): the pillar canvas's occupancy map [B,1,H,W] -- the first layer then runs its sparse form"""
        fold, taps = None, []
        for i, block in enumerate(self.blocks):
            mods = list(block)
            st = mods[1].stride[0]
            x, fold = MC.fused_conv(x, fold, mods[1], out_bn=mods[2], spec=MC.ConvSpec(3, 3, st, 1),  # ZeroPad2d(1) + conv(pad 0)
                                    occupancy=occupancy if (i == 0 and fold is None) else None)
            for j in range(4, len(mods), 3):
                x, fold = MC.fused_conv(x, fold, mods[j], out_bn=mods[j + 1])
            cut = getattr(self, "grad_cut", None)
            if cut is not None and i == 0:  # (extension) behind block 0, in front of its two consumers: see mfma_conv.GradCut
                x = cut.split(x)
            if i - self._upsample_start_idx >= 0:
                taps.append((x, fold, self.deblocks[i - self._upsample_start_idx]))
        if len(taps) == 0:
            return x, fold
        if len(taps) == 1:
            xi, fi, d = taps[0]
            return MC.fused_conv(xi, fi, d[0], out_bn=d[1])
        # the up-sampling branches write their raw maps straight into the channel ranges of ONE buffer (reference :140-146:
        # torch.cat(ups, dim=1)): no concatenation pass forward, channel-slice views of the gradient backward
        specs = [MC.ConvSpec.of(d[0]) for _, _, d in taps]
        sizes = [s_.out_hw(xi.shape[2], xi.shape[3]) for s_, (xi, _, _) in zip(specs, taps)]
        chans = [d[0].out_channels for _, _, d in taps]
        vec = 8 if x.dtype == torch.bfloat16 else 4
        ups = []
        if len(set(sizes)) == 1 and all(c % 8 == 0 for c in chans) and sum(chans) % vec == 0:
            (ho, wo), B = sizes[0], x.shape[0]
            buf = torch.empty((B, ho, wo, sum(chans)), dtype=x.dtype, device=x.device)
            off = 0
            for (xi, fi, d), c in zip(taps, chans):
                ups.append(MC.fused_conv(xi, fi, d[0], out_bn=d[1], out=(buf, off)))
                off += c
            raw = MC.slice_cat(buf.permute(0, 3, 1, 2), [u[0] for u in ups])
        else:
            ups = [MC.fused_conv(xi, fi, d[0], out_bn=d[1]) for xi, fi, d in taps]
            raw = torch.cat([u[0] for u in ups], dim=1)
        return raw, MC.BnFold.cat([u[1] for u in ups])
