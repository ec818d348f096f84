import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.networks.centerpoint.fused_bn import bn_act
torch.manual_seed(0)
for (n, c, h, w) in ((1, 64, 256, 256), (1, 128, 128, 128), (1, 256, 64, 64), (1, 64, 128, 128), (4, 64, 256, 256), (1, 384, 1, 1) if False else (1, 64, 37, 53)):
    x = (torch.randn(n, c, h, w, device="cuda") * 0.7 + 30.0).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(c).cuda()
    y = bn_act(x, bn, relu=True)
    g = torch.randn_like(y)
    y.backward(g)
    xd = x.detach().double()
    mean = xd.mean(dim=(0, 2, 3)); var = xd.var(dim=(0, 2, 3), unbiased=False)
    yr = torch.relu((xd - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + bn.eps))
    print((n, c, h, w), "fwd err %.3e" % float((y.double() - yr).abs().max()), "rm err %.3e" % float((bn.running_mean.double() - 0.1 * mean).abs().max()),
          "gsum %.10f" % float(x.grad.double().abs().sum()), "ggamma %.10f" % float(bn.weight.grad.double().abs().sum()))
