import torch, torch.nn.functional as F
torch.manual_seed(0)
def rel(a,b): return ((a.double().cpu()-b).abs().max()/b.abs().max()).item()
def test(name, x, cl=True):
    C = x.shape[1]
    gam = torch.rand(C)+0.5; bet = torch.randn(C)
    xd = x.double().requires_grad_(True)
    yd = F.relu(F.batch_norm(xd, None, None, gam.double(), bet.double(), True, 0.1, 1e-5)); g = torch.randn_like(yd); (yd*g).sum().backward()
    for dev in ('cpu','cuda'):
        xg = x.to(dev)
        if cl: xg = xg.contiguous(memory_format=torch.channels_last)
        xg.requires_grad_(True)
        yg = F.relu(F.batch_norm(xg, None, None, gam.to(dev), bet.to(dev), True, 0.1, 1e-5)); (yg*g.float().to(dev)).sum().backward()
        print(name, dev, 'fwd %.1e bwd %.1e' % (rel(yg, yd.detach()), rel(xg.grad, xd.grad)))
test('randn', torch.randn(2,64,64,64))
test('offset5 std0.1', torch.randn(2,64,64,64)*0.1+5)
test('offset50 std0.1', torch.randn(2,64,64,64)*0.1+50)
x = torch.randn(2,64,64,64) * (torch.rand(2,1,64,64)>0.9); test('sparse 10%', x)
x = (torch.randn(2,64,64,64)+3) * (torch.rand(2,1,64,64)>0.98); test('sparse 2% offset', x)
test('offset5 std0.1 nchw', torch.randn(2,64,64,64)*0.1+5, cl=False)
