import os, sys
sys.path.insert(0, "tests")
import numpy as np, torch
import test_gpu_slim as T
g = T._g()
fnet, cnet, ub, raft, dec = T._build()
gi = torch.Generator().manual_seed(5)
img0 = torch.randn(1, 64, 128, 128, generator=gi) * (torch.rand(1, 1, 128, 128, generator=gi) > 0.8)
img1 = torch.roll(img0, shifts=(3, -2), dims=(2, 3)) + 0.05 * torch.randn(1, 64, 128, 128, generator=gi)
for m in (fnet, cnet, ub):
    m.cuda()
img0, img1 = img0.cuda(), img1.cuda()
fmap0, fmap1 = fnet(img0), fnet(img1)
preds = raft.predict_single_flow_map_and_classes(img0, fmap0, fmap1, dec)
wts = [torch.randn(preds[0].shape, generator=gi).cuda() for _ in preds]
sum((p * wt).sum() for p, wt in zip(preds, wts)).backward()
def st(a, b, name):
    a = a.detach().double().cpu().numpy(); d = np.abs(a - b)
    print(f"{name:22s} max {d.max()/np.abs(b).max():.2e} median {np.median(d)/np.median(np.abs(b)):.2e} frac>1e-2max {(d > 1e-2*np.abs(b).max()).mean():.3f}")
print("backend", os.environ.get("LISO_CONV_BACKEND", "mfma"))
st(fmap0, g["raft_fmap0"], "fmap0")
st(preds[-1][:, ::2, ::2], g["raft_pred_last"], "pred_last")
st(fnet.conv1.weight.grad, g["raft_g_fnet_conv1"], "g fnet.conv1")
st(cnet.conv2.weight.grad, g["raft_g_cnet_conv2"], "g cnet.conv2")
st(ub.gru.convz.weight.grad[:, ::8], g["raft_g_gru_convz"], "g gru.convz")
st(ub.static_flow_head.conv2.weight.grad, g["raft_g_flow_head"], "g flow_head")
st(ub.motion_encoder.conv_stat_corr1.weight.grad[..., 0, 0], g["raft_g_corr_conv"], "g corr conv")
