import torch, torch.nn.functional as F
torch.manual_seed(0)
def rel(a,b): return ((a.double().cpu()-b).abs().max()/b.abs().max()).item()
def test(name, fn, x, *params, cl=True):
    xd = x.double().requires_grad_(True); pd = [p.double().requires_grad_(True) for p in params]
    yd = fn(xd, *pd); g = torch.randn_like(yd); (yd*g).sum().backward()
    xg = x.cuda()
    if cl and x.dim()==4: xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True); pg = [p.cuda().requires_grad_(True) for p in params]
    yg = fn(xg, *pg); (yg*g.float().cuda()).sum().backward()
    print(name, 'fwd %.1e' % rel(yg, yd.detach()), 'dx %.1e' % rel(xg.grad, xd.grad), ' '.join('dp%d %.1e' % (i, rel(a.grad, b.grad)) for i,(a,b) in enumerate(zip(pg,pd))))
for co in (1,2,3):
    test(f'conv3x3 64->{co} bias 32x32', lambda x,w,b: F.conv2d(x,w,b,padding=1), torch.randn(2,64,32,32), torch.randn(co,64,3,3)*0.05, torch.randn(co))
test('conv3x3 384->64 bias 32x32', lambda x,w,b: F.conv2d(x,w,b,padding=1), torch.randn(2,384,32,32), torch.randn(64,384,3,3)*0.02, torch.randn(64))
test('conv3x3 64->64 bias 32x32', lambda x,w,b: F.conv2d(x,w,b,padding=1), torch.randn(2,64,32,32), torch.randn(64,64,3,3)*0.05, torch.randn(64))
test('convT 2x2s2 256->128 4x4', lambda x,w: F.conv_transpose2d(x,w,None,stride=2), torch.randn(2,256,16,16), torch.randn(256,128,2,2)*0.05)
test('conv1x1 128->128 32x32', lambda x,w: F.conv2d(x,w,None,stride=1), torch.randn(2,128,32,32), torch.randn(128,128,1,1)*0.05)
test('conv2x2s2 64->128 64x64', lambda x,w: F.conv2d(x,w,None,stride=2), torch.randn(2,64,64,64), torch.randn(128,64,2,2)*0.05)
test('conv3x3s2 64->64 pad1 128x128', lambda x,w: F.conv2d(x,w,None,stride=2,padding=1), torch.randn(2,64,128,128), torch.randn(64,64,3,3)*0.05)
test('conv3x3 256->256 16x16', lambda x,w: F.conv2d(x,w,None,padding=1), torch.randn(2,256,16,16), torch.randn(256,256,3,3)*0.02)
test('conv3x3 128->128 32x32', lambda x,w: F.conv2d(x,w,None,padding=1), torch.randn(2,128,32,32), torch.randn(128,128,3,3)*0.02)
def catfn(x): 
    a,b,c = x[:, :128], x[:,128:256], x[:,256:]
    return torch.cat([F.relu(a), F.relu(b), F.relu(c)], 1)
test('cat', catfn, torch.randn(2,384,32,32))
test('bn small', lambda x,g,b: F.relu(F.batch_norm(x,None,None,g,b,True,0.1,1e-5)), torch.randn(2,256,16,16), torch.rand(256)+0.5, torch.randn(256))
