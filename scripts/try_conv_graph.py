import torch
from liso_amd.utils import mfma_conv as MC
dev = torch.device("cuda")
torch.manual_seed(0)
spec = MC.ConvSpec(3, 3, 1, 1)
x = torch.randn(2, 64, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
dy = torch.randn(2, 64, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

def run():
    y, part = MC.conv_forward(x, w, None, spec, want_stats=True)
    dx = MC.conv_dgrad(dy, w, spec, tuple(x.shape))
    dw, db = MC.conv_wgrad(x, dy, tuple(w.shape), spec)
    return y, part, dx, dw

ref = [t.clone() for t in run()]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
for rep in range(2):
    g.replay()
    torch.cuda.synchronize()
    for name, a, b in zip(("y", "stats", "dx", "dw"), out, ref):
        print(rep, name, float((a.float() - b.float()).abs().max()), float(b.float().abs().max()))
# fused chain
conv1 = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
bn1 = torch.nn.BatchNorm2d(64).to(dev)
conv2 = torch.nn.Conv2d(64, 64, 3, padding=1, bias=True).to(dev)
xin = torch.randn(2, 64, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
def chain():
    for p in list(conv1.parameters()) + list(bn1.parameters()) + list(conv2.parameters()):
        p.grad = None
    y, f = MC.fused_conv(xin, None, conv1, out_bn=bn1)
    y, _ = MC.fused_conv(y, f, conv2)
    loss = (y * y).mean()
    loss.backward()
    return loss.detach(), conv1.weight.grad.clone(), bn1.weight.grad.clone()
ref = [t.clone() for t in chain()]
rm = bn1.running_mean.clone()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain()
torch.cuda.current_stream().wait_stream(s)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    out = chain()
g2.replay(); torch.cuda.synchronize()
for name, a, b in zip(("loss", "gw1", "ggamma"), out, ref):
    print("chain", name, float((a.float() - b.float()).abs().max()), float(b.float().abs().max()))
