"""stage-by-stage errors of the fused conv/BN chain against fp64 torch modules"""
import copy
import torch
from liso_amd.utils import mfma_conv as MC

torch.manual_seed(0)
convs = [torch.nn.Conv2d(32, 64, 3, stride=2, padding=1, bias=False), torch.nn.Conv2d(64, 64, 3, padding=1, bias=True),
         torch.nn.Conv2d(64, 8, 3, padding=1, bias=True)]
bns = [torch.nn.BatchNorm2d(64), torch.nn.BatchNorm2d(64)]
with torch.no_grad():
    for bn in bns:
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.3, 0.3)
x = torch.randn(2, 32, 48, 40)
ref = copy.deepcopy(torch.nn.Sequential(convs[0], bns[0], torch.nn.ReLU(), convs[1], bns[1], torch.nn.ReLU(), convs[2])).double().train()
x64 = x.double().requires_grad_(True)
acts = []
h = x64
for m in ref:
    h = m(h)
    h.retain_grad()
    acts.append(h)
out64 = h
wgt = torch.linspace(-1, 1, out64.numel()).view_as(out64).double()
(out64 * wgt).sum().backward()
for m in convs + bns:
    m.cuda().train()


def rel(a, b, name):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs()
    print(f"{name:28s} max {float(d.max() / b.abs().max()):.2e}  median {float(d.median() / b.abs().median().clamp(min=1e-30)):.2e}  "
          f"frac>1e-3*max {float((d > 1e-3 * b.abs().max()).double().mean()):.2e}")


xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
y0, f0 = MC.fused_conv(xd, None, convs[0], out_bn=bns[0])
y0.retain_grad()
rel(y0, acts[0], "conv0 out")
C = 64
n0 = (y0.float() * f0.stats[:C].view(1, C, 1, 1) + f0.stats[C:2 * C].view(1, C, 1, 1)).relu()
rel(n0, acts[2], "bn0+relu (from fold)")
y1, f1 = MC.fused_conv(y0, f0, convs[1], out_bn=bns[1])
y1.retain_grad()
rel(y1, acts[3], "conv1 out")
y2, _ = MC.fused_conv(y1, f1, convs[2])
rel(y2, acts[6], "conv2 out")
(y2 * wgt.float().cuda()).sum().backward()
rel(y1.grad, acts[3].grad, "grad conv1 out (after bn1 bwd)")
rel(y0.grad, acts[0].grad, "grad conv0 out (after bn0 bwd)")
rel(xd.grad, x64.grad, "grad x")
for i, j in ((0, 0), (1, 3), (2, 6)):
    rel(convs[i].weight.grad, ref[j].weight.grad, f"conv{i}.weight.grad")
rel(convs[2].bias.grad, ref[6].bias.grad, "conv2.bias.grad")
for i, j in ((0, 1), (1, 4)):
    rel(bns[i].weight.grad, ref[j].weight.grad, f"bn{i}.gamma.grad")
    rel(bns[i].bias.grad, ref[j].bias.grad, f"bn{i}.beta.grad")
# the existing torch path for comparison (fp32 torch modules on the GPU)
ref32 = copy.deepcopy(ref).float().cuda().train()
x32 = x.cuda().requires_grad_(True)
o = ref32(x32)
(o * wgt.float().cuda()).sum().backward()
rel(x32.grad, x64.grad, "torch fp32 GPU: grad x")
rel(ref32[0].weight.grad, ref[0].weight.grad, "torch fp32 GPU: conv0.w.grad")
