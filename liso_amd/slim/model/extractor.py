"""SLIM feature encoders.  Mirror of liso/slim/model/extractor.py:5-71,211-297 (ResidualBlock, SmallEncoder; same
constructor arguments and attribute names -> same state_dict keys).  BottleneckBlock / BasicEncoder of the reference
are not used by SLIM's RAFT and are out of scope."""
import torch
import torch.nn as nn

from liso_amd.slim.model.fused_norm import in_act
from liso_amd.utils.mfma_conv import conv2d


def _norm(norm_fn, ch, groups=None):
    if norm_fn == "group":
        return nn.GroupNorm(num_groups=groups, num_channels=ch)
    if norm_fn == "batch":
        return nn.BatchNorm2d(ch)
    if norm_fn == "instance":
        return nn.InstanceNorm2d(ch)
    if norm_fn == "instance_affine":
        return nn.InstanceNorm2d(ch, eps=1e-3, affine=True)
    if norm_fn == "none":
        return nn.Sequential()
    raise ValueError(norm_fn)


class ResidualBlock(nn.Module):
    def __init__(self, in_filters, out_filters, dummy_in_filters, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_filters, out_filters, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(out_filters, out_filters, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        needs_down = not (stride == 1 and dummy_in_filters == out_filters)  # reference :19-21
        self.norm1 = _norm(norm_fn, out_filters, out_filters // 8)
        self.norm2 = _norm(norm_fn, out_filters, out_filters // 8)
        if needs_down:
            self.norm3 = _norm(norm_fn, out_filters, out_filters // 8)
            self.downsample = nn.Sequential(nn.Conv2d(in_filters, out_filters, kernel_size=1, stride=stride), self.norm3)
        else:
            self.downsample = None

    def forward_inference(self, x_raw, fold):
        """forward() under no_grad on fp32 CUDA tensors with the normalisations folded: (raw block input, its pending InstanceNorm or
        None) -> materialised block output.  conv1 -> [IN+ReLU pending] -> conv2 (prologue) -> [IN+ReLU pending] -> one residual
        kernel that applies the pending layers of both branches: no normalised tensor, no separate ReLU / add pass, NHWC throughout."""
        from liso_amd.utils import mfma_conv as MC

        y, f1 = MC.conv_in(x_raw, fold, self.conv1, self.norm1)
        y, f2 = MC.conv_in(y, f1, self.conv2, self.norm2)
        if self.downsample is not None:
            xs, fx = MC.conv_in(x_raw, fold, self.downsample[0], self.downsample[1], relu=False)
        else:
            xs, fx = x_raw, fold
        return MC.residual_relu(xs, fx, y, f2)

    def foldable(self):
        from liso_amd.utils import mfma_conv as MC

        norms = [self.norm1, self.norm2] + ([self.downsample[1]] if self.downsample is not None else [])
        kinds = {MC._norm_kind(n) for n in norms}
        return len(kinds) == 1 and None not in kinds

    def forward(self, x):
        """reference :29-38.  Convolutions on the own MFMA kernels (liso_amd/utils/mfma_conv.py); where no normalisation
        sits between a convolution and its ReLU (cnet: norm_fn "none") the ReLU runs in the convolution's epilogue."""
        bare = isinstance(self.norm1, nn.Sequential) and len(self.norm1) == 0
        if bare:
            y = conv2d(self.conv2, conv2d(self.conv1, x, relu=True), relu=True)
        else:  # (InstanceNorm + ReLU: one set of channels-last passes, fused_norm.py; other norms: the modules)
            y = in_act(conv2d(self.conv1, x), self.norm1)
            y = in_act(conv2d(self.conv2, y), self.norm2)
        if self.downsample is not None:
            x = in_act(conv2d(self.downsample[0], x), self.downsample[1], relu=False)
        from liso_amd.utils.mfma_conv import add_relu

        return add_relu(x, y)  # (one launch on the GPU; torch.relu(x + y) elsewhere)


class SmallEncoder(nn.Module):
    """conv7x7/2 64->32, three stages of two residual blocks (32, 64 /2, 96 /2), conv1x1 -> output_dim at 1/8 res."""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 32, 8)
        self.conv1 = nn.Conv2d(in_channels=64, out_channels=32, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(32, 32, stride=1)
        self.layer2 = self._make_layer(32, 64, stride=2)
        self.layer3 = self._make_layer(64, 96, stride=2)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        self.conv2 = nn.Conv2d(96, output_dim, kernel_size=1)

    def _make_layer(self, in_filters, out_filters, stride=1):
        # reference :258-276: the second block's "dummy_in_filters" is the stage input width, so it gets a 1x1
        # projection whenever the stage changes the width
        return nn.Sequential(
            ResidualBlock(in_filters, out_filters, norm_fn=self.norm_fn, dummy_in_filters=in_filters, stride=stride),
            ResidualBlock(out_filters, out_filters, norm_fn=self.norm_fn, dummy_in_filters=in_filters, stride=1))

    def _fold_inference(self, x):
        from liso_amd.utils import mfma_conv as MC

        if torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32:
            return False
        if not getattr(self, "fold_inference", True) or MC._norm_kind(self.norm1) is None:
            return False
        blocks = list(self.layer1) + list(self.layer2) + list(self.layer3)
        return all(b.foldable() and MC._norm_kind(b.norm1) == MC._norm_kind(self.norm1) for b in blocks) and \
            MC.supported(x, self.conv1.weight, MC.ConvSpec.of(self.conv1))

    def forward(self, x, occupancy=None):
        """`occupancy` (extension): the occupancy map of the pillar canvas `x` (fp32 [B,1,H,W], 0 = no pillar): the first
        convolution then skips the tiles of the (sparse) BEV canvas that hold no pillar at all -- same result, bit for bit -- and,
        in training, its weight gradient walks the occupied cells only (mfma_conv.conv_wgrad_sparse)"""
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        if self._fold_inference(x):
            # inference: every InstanceNorm + ReLU is applied by its consumer, residual tails are one kernel (mfma_conv.InFold)
            from liso_amd.utils import mfma_conv as MC

            h, fold = MC.conv_in(x, None, self.conv1, self.norm1, occupancy=occupancy if not is_list else None)
            for blk in list(self.layer1) + list(self.layer2) + list(self.layer3):
                h, fold = blk.forward_inference(h, fold), None
            x = conv2d(self.conv2, h)
        else:
            occ = occupancy if (occupancy is not None and not is_list and x.is_cuda) else None
            if isinstance(self.norm1, nn.Sequential) and len(self.norm1) == 0:
                x = conv2d(self.conv1, x, relu=True, occupancy=occ)
            else:
                x = in_act(conv2d(self.conv1, x, occupancy=occ), self.norm1)
            x = self.layer3(self.layer2(self.layer1(x)))
            x = conv2d(self.conv2, x)
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        if is_list:
            x = torch.split(x, [batch_dim, batch_dim], dim=0)
        return x
