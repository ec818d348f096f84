import torch, torch.nn.functional as F
torch.manual_seed(0)
def rel(a,b): return ((a.double().cpu()-b).abs().max()/b.abs().max()).item()
for (cin,cout,hw,stride,cl) in [(64,64,64,1,True),(64,64,64,1,False),(128,128,32,1,True),(384,64,32,1,True),(64,64,64,2,True)]:
    x = torch.randn(2,cin,hw,hw); w = torch.randn(cout,cin,3,3)*0.05
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, stride=stride, padding=1); g = torch.randn_like(yd)
    (yd*g).sum().backward()
    xg = x.cuda(); 
    if cl: xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True); wg = w.cuda().requires_grad_(True)
    yg = F.conv2d(xg, wg, None, stride=stride, padding=1)
    (yg*g.float().cuda()).sum().backward()
    print((cin,cout,hw,stride,'NHWC' if cl else 'NCHW'), 'fwd', rel(yg, yd.detach()), 'dgrad', rel(xg.grad, xd.grad), 'wgrad', rel(wg.grad, wd.grad))
# BN backward
x = torch.randn(2,64,64,64); xd = x.double().requires_grad_(True)
gam = torch.rand(64)+0.5; bet = torch.randn(64)
yd = F.relu(F.batch_norm(xd, None, None, gam.double(), bet.double(), True, 0.1, 1e-5)); g = torch.randn_like(yd); (yd*g).sum().backward()
xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
yg = F.relu(F.batch_norm(xg, None, None, gam.cuda(), bet.cuda(), True, 0.1, 1e-5), inplace=True); (yg*g.float().cuda()).sum().backward()
print('bn+relu fwd', rel(yg, yd.detach()), 'bwd', rel(xg.grad, xd.grad))
