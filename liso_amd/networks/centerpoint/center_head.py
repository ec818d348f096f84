"""CenterHead / SepHead.  Mirror of liso/networks/centerpoint/center_head.py (same ctor arguments and module names ->
state_dict keys `shared_conv.*`, `tasks.0.{pos,dims,rot,probs}.*`)."""
import copy
from typing import Dict

import torch
from torch import nn

from liso_amd.networks.centerpoint.rpn import conv_bn_relu
from liso_amd.networks.centerpoint.weight_init import kaiming_init


class _BlockDiagonalFilters(torch.autograd.Function):
    """The output convolutions of the heads -- same geometry, each on its own 64-channel slice of the merged hidden map -- as ONE
    convolution: filters [sum co_k, sum ci_k, kh, kw] with head k's filter in its diagonal block and zeros elsewhere, biases
    concatenated.  One forward, one data-gradient and one weight-gradient launch instead of four of each on 1-3 output channels
    (launch-bound: 9 + 21 + 28 us per head at B = 4).  Backward: the diagonal blocks of the dense weight gradient."""

    @staticmethod
    def forward(ctx, n, *wb):
        ws, bs = wb[:n], wb[n:]
        co = [w.shape[0] for w in ws]
        ci = [w.shape[1] for w in ws]
        W = ws[0].new_zeros((sum(co), sum(ci)) + tuple(ws[0].shape[2:]))
        ctx.co, ctx.ci = co, ci
        ctx.device_blocks = W.is_cuda and W.dtype == torch.float32
        if ctx.device_blocks:  # every block and bias by one launch (include/liso_optim.h: liso_multi_copy_rows)
            from liso_amd import _lib as L

            bias = bs[0].new_empty(sum(co))
            jobs, o, c = [], 0, 0
            for w, b_, a, b in zip(ws, bs, co, ci):
                jobs += [(W[o:o + a, c:c + b], w.detach()), (bias[o:o + a], b_.detach())]
                o, c = o + a, c + b
            L.copy_blocks(jobs)
            return W, bias
        o = c = 0
        for w, a, b in zip(ws, co, ci):
            W[o:o + a, c:c + b] = w
            o, c = o + a, c + b
        return W, torch.cat(bs)

    @staticmethod
    def backward(ctx, gW, gb):
        gws, gbs, o, c = [], [], 0, 0
        if ctx.device_blocks and gW is not None and gW.dtype == torch.float32:
            from liso_amd import _lib as L

            jobs = []
            for a, b in zip(ctx.co, ctx.ci):
                gws.append(gW.new_empty((a, b) + tuple(gW.shape[2:])))
                jobs.append((gws[-1], gW[o:o + a, c:c + b]))
                gbs.append(gb[o:o + a] if gb is not None else None)
                o, c = o + a, c + b
            L.copy_blocks(jobs)
            return (None, *gws, *gbs)
        for a, b in zip(ctx.co, ctx.ci):
            gws.append(gW[o:o + a, c:c + b].contiguous() if gW is not None else None)
            gbs.append(gb[o:o + a] if gb is not None else None)
            o, c = o + a, c + b
        return (None, *gws, *gbs)


class SepHead(nn.Module):
    def __init__(self, in_channels, heads, norm_cfg, head_conv=64, final_kernel=1, bn=False, **kwargs):
        super().__init__(**kwargs)
        self.heads = heads
        for head in self.heads:
            classes, num_conv = self.heads[head]
            fc = []
            for _ in range(num_conv - 1):  # reference :28-42
                fc.append(nn.Conv2d(in_channels, head_conv, kernel_size=final_kernel, stride=1,
                                    padding=final_kernel // 2, bias=True))
                if bn:
                    fc.append(nn.BatchNorm2d(head_conv, **norm_cfg))
                fc.append(nn.ReLU())
            fc.append(nn.Conv2d(head_conv, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2,
                                bias=True))
            fc = nn.Sequential(*fc)
            for m in fc.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
            self.__setattr__(head, fc)

    def forward_fused(self, x_raw, fold, MC):
        """all heads' hidden convolutions (same input, same geometry) as ONE convolution with 4 x head_conv filters and one
        statistics epilogue; every output convolution then reads its 64-channel slice of that map (fp32 outputs)"""
        seqs = [list(self.__getattr__(h)) for h in self.heads]
        if not all(len(q) == 4 and isinstance(q[1], nn.BatchNorm2d) for q in seqs):
            return None
        hid, hfold = MC.fused_conv(x_raw, fold, [q[0] for q in seqs], out_bn=[q[1] for q in seqs])
        lasts = [q[3] for q in seqs]
        same = all(l.kernel_size == lasts[0].kernel_size and l.stride == lasts[0].stride and l.padding == lasts[0].padding and
                   l.bias is not None for l in lasts)
        if same and getattr(self, "merge_output_convs", True):
            # the heads' output convolutions as one block-diagonal convolution on the whole hidden map (see _BlockDiagonalFilters);
            # every head reads its channels of the result (a split: its backward is one concatenation of the heads' gradients)
            import types

            W, b = _BlockDiagonalFilters.apply(len(lasts), *[l.weight for l in lasts], *[l.bias for l in lasts])
            y, _ = MC.fused_conv(hid, hfold, types.SimpleNamespace(weight=W, bias=b), out_dtype=torch.float32, spec=MC.ConvSpec.of(lasts[0]))
            return dict(zip(self.heads, torch.split(y, [l.out_channels for l in lasts], dim=1)))
        # (split, not four slices: the backward of split is ONE concatenation of the heads' input gradients, the backward of each
        # slice a zero-filled full-width map + a copy, and autograd then adds the four maps)
        parts = torch.split(hid, [q[0].out_channels for q in seqs], dim=1)
        out = {}
        for k, (head, q) in enumerate(zip(self.heads, seqs)):
            out[head], _ = MC.fused_conv(parts[k], hfold.group(k), q[3], out_dtype=torch.float32)
        return out

    def forward(self, x):
        """host tensors (CPU test tier); device tensors take `forward_fused`"""
        from liso_amd.utils import host_ops

        out = {}
        for head in self.heads:
            fc = list(self.__getattr__(head))
            y = x
            i = 0
            while i < len(fc) - 1:  # conv (+BN) + ReLU groups
                if isinstance(fc[i + 1], nn.BatchNorm2d):
                    y = conv_bn_relu(y, fc[i], fc[i + 1])
                    i += 3
                else:
                    y = torch.relu(host_ops.conv2d(y, fc[i].weight.to(y.dtype), fc[i].bias.to(y.dtype), padding=fc[i].padding))
                    i += 2
            last = fc[-1]
            # final prediction conv: fp32 output (logits / regression targets feed an fp32 loss)
            out[head] = host_ops.conv2d(y, last.weight.to(y.dtype), last.bias.to(y.dtype), padding=last.padding).float()
        return out


class CenterHead(nn.Module):
    def __init__(self, common_heads: Dict, norm_cfg, in_channels=(128,), stride=1, share_conv_channel=64):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = 1
        self.shared_conv = nn.Sequential(  # reference :81-92
            nn.Conv2d(in_channels, share_conv_channel, stride=stride, kernel_size=3, padding=1, bias=True),
            nn.BatchNorm2d(share_conv_channel, **norm_cfg), nn.ReLU(inplace=True))
        self.tasks = nn.ModuleList()
        self.tasks.append(SepHead(share_conv_channel, copy.deepcopy(common_heads), norm_cfg=norm_cfg, bn=True,
                                  final_kernel=3))

    def forward(self, x, *kwargs):
        from liso_amd.utils import mfma_conv as MC

        fold = None
        if isinstance(x, tuple):  # (raw maps, BnFold) from RPN.forward(lazy=True)
            x, fold = x
        if MC.on_device(x):
            s_raw, s_fold = MC.fused_conv(x, fold, self.shared_conv[0], out_bn=self.shared_conv[1])
            assert len(self.tasks) == 1, len(self.tasks)
            ret = self.tasks[0].forward_fused(s_raw, s_fold, MC)
            if ret is not None:
                return ret
            x = MC.materialize(s_raw, s_fold)
            return self.tasks[0](x)
        x = MC.materialize(x, fold)
        x = conv_bn_relu(x, self.shared_conv[0], self.shared_conv[1])
        ret = [task(x) for task in self.tasks]
        assert len(ret) == 1, len(ret)
        return ret[0]
