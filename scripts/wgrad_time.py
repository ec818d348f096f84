"""conv_wgrad (+ slab reduction) of one 3x3 / stride-1 layer, event-timed over 50 calls after warm-up; checks against an fp64 reference.
python scripts/wgrad_time.py B ci co H [fp32|bf16]      (LISO_WGRAD_CIW = 64 | 32: input channels per block of conv_wgrad_rs3_kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from liso_amd.utils import mfma_conv as MC  # noqa: E402

B, ci, co, H = [int(v) for v in sys.argv[1:5]]
dt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == "fp32") else torch.bfloat16
spec = MC.ConvSpec(3, 3, 1, 1, False)
torch.manual_seed(0)
x = torch.randn(B, ci, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, co, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
for _ in range(5):
    dw, db = MC.conv_wgrad(x, dy, (co, ci, 3, 3), spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    MC.conv_wgrad(x, dy, (co, ci, 3, 3), spec)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
xr = x.double().requires_grad_(False)
w = torch.zeros(co, ci, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
b = torch.zeros(co, device="cuda", dtype=torch.float64, requires_grad=True)
y = torch.nn.functional.conv2d(xr, w, b, padding=1)
y.backward(dy.double())
rel = float((dw.double() - w.grad).abs().max() / w.grad.abs().max())
relb = float((db.double() - b.grad).abs().max() / b.grad.abs().max())
print(f"B{B} {ci}->{co} @{H} {str(dt).split('.')[-1]}: {us:7.1f} us per call (eager, wgrad + reduce)   max err / max |dw| {rel:.2e}  bias {relb:.2e}")
