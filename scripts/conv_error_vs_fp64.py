"""forward error of the F32X3 convolution kernels against an fp64 reference on one layer: python scripts/conv_error_vs_fp64.py B ci co H"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.utils import mfma_conv as MC
B, ci, co, H = [int(v) for v in sys.argv[1:5]]
torch.manual_seed(0)
spec = MC.ConvSpec(3, 3, 1, 1, False)
x = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
y, _ = MC.conv_forward(x, w, None, spec)
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
err = (y.double() - ref).abs()
print(f"LISO_CONV_ROLES={os.environ.get('LISO_CONV_ROLES', '1')}: B{B} {ci}->{co} @{H}: max |err| / max |ref| = {float(err.max() / ref.abs().max()):.3e}, "
      f"rms err / rms ref = {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")
