import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.nn.functional as F
import test_gpu_detector as T
from liso_amd.networks.centerpoint.rpn import conv_bn_relu
tr, pcls, targets = T._setup(128, 100.0, 2, 20000)
head = tr.net.model.center_head
cap = {}
def fwd(x, *a):
    cap["x_in"] = x.detach().clone()
    x1 = conv_bn_relu(x, head.shared_conv[0], head.shared_conv[1])
    x1.retain_grad(); cap["x1"] = x1
    return head.tasks[0](x1)
head.forward = fwd
tr.model.train()
rm0 = head.shared_conv[1].running_mean.clone()
total, _, _ = tr.loss(pcls, targets)
total.backward()
g1 = cap["x1"].grad.detach().double().cpu()
x_in = cap["x_in"].double().cpu().requires_grad_(True)
w = head.shared_conv[0].weight.detach().double().cpu().requires_grad_(True)
b = head.shared_conv[0].bias.detach().double().cpu().requires_grad_(True)
gam = head.shared_conv[1].weight.detach().double().cpu().requires_grad_(True)
bet = head.shared_conv[1].bias.detach().double().cpu().requires_grad_(True)
y = F.relu(F.batch_norm(F.conv2d(x_in, w, b, padding=1), None, None, gam, bet, True, 0.1, head.shared_conv[1].eps))
print("x1 fwd err", float((y.detach() - cap["x1"].detach().double().cpu()).abs().max()), "x1 absmax", float(y.abs().max()))
y.backward(g1)
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
print("given the GPU's own dL/dx1: conv.weight %.2e  bn.weight %.2e  bn.bias %.2e" % (
    rel(head.shared_conv[0].weight.grad, w.grad), rel(head.shared_conv[1].weight.grad, gam.grad), rel(head.shared_conv[1].bias.grad, bet.grad)))
print("g1 stats: absmax %.3e, nonzero frac %.3f, dtype %s, strides %s" % (float(g1.abs().max()), float((g1 != 0).double().mean()), cap["x1"].grad.dtype, cap["x1"].grad.stride()))

# ---- the oracle's dL/dx1 -----------------------------------------------------------------------------------------------
import oracle.detector as OD
from oracle.train_step import detector_forward_loss, prepare_state
ocap = {}
def chf(sd, x, heads, training, prefix=""):
    p = prefix + "shared_conv."
    x = F.relu(OD._bn(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1), sd, p + "1", training))
    x.retain_grad(); ocap["x1"] = x
    out = {}
    for h in heads:
        q = f"{prefix}tasks.0.{h}."
        z = F.conv2d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=1)
        z.retain_grad(); ocap["z_" + h] = z
        y = F.relu(OD._bn(z, sd, q + "1", training))
        out[h] = F.conv2d(y, sd[q + "3.weight"], sd[q + "3.bias"], padding=1)
    return out
OD.center_head_forward = chf
import oracle.train_step as OT
if hasattr(OT, "center_head_forward"): OT.center_head_forward = chf
sd0 = {k: v for k, v in tr.net.state_dict().items()}
sd0["model.center_head.shared_conv.1.running_mean"] = rm0  # irrelevant in training mode
sd64 = prepare_state(sd0, torch.float64)
ref64, _, _ = detector_forward_loss(sd64, [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}, 128, 100.0, dtype=torch.float64)
ref64.backward()
go = ocap["x1"].grad
print("x1: gpu vs oracle fwd %.2e" % rel(cap["x1"].detach(), ocap["x1"].detach()))
print("dL/dx1: gpu vs oracle %.3e   (sum per channel: gpu %.4e oracle %.4e)" % (rel(g1, go), float(g1.sum((0, 2, 3)).abs().max()), float(go.sum((0, 2, 3)).abs().max())))
d = (g1 - go)
print("diff absmax %.3e at border rows/cols? interior absmax %.3e" % (float(d.abs().max()), float(d[:, :, 1:-1, 1:-1].abs().max())))
print("oracle shared bn.bias grad vs gpu", rel(head.shared_conv[1].bias.grad, sd64["model.center_head.shared_conv.1.bias"].grad))
chk = (go * (ocap["x1"] > 0)).sum((0, 2, 3))
ob = sd64["model.center_head.shared_conv.1.bias"].grad
gb = head.shared_conv[1].bias.grad.double().cpu()
print("oracle leaf grad vs sum(go*mask): %.3e ; gpu vs sum(go*mask): %.3e" % (float((ob - chk).abs().max() / chk.abs().max()), float((gb - chk).abs().max() / chk.abs().max())))
print("ob[:4]", ob[:4].tolist(), "\ngb[:4]", gb[:4].tolist(), "\nchk[:4]", chk[:4].tolist())
print("is leaf shared with other keys?", [k for k, v in sd64.items() if v is sd64["model.center_head.shared_conv.1.bias"]])
