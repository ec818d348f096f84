"""run one forward convolution shape repeatedly (for rocprofv3 --pmc)"""
import sys
import torch
from liso_amd.utils import mfma_conv as MC
ci, co, H, k, s, p = (int(v) for v in sys.argv[1:7])
kind = sys.argv[7] if len(sys.argv) > 7 else "fwd"
B = 4
spec = MC.ConvSpec(k, k, s, p, False)
x = torch.randn(B, ci, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, k, k, device="cuda") * 0.05
ho, wo = spec.out_hw(H, H)
dy = torch.randn(B, co, ho, wo, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
packed = MC.pack_weights(w, spec, False, 0)
for _ in range(10):
    if kind == "fwd":
        MC.conv_forward(x, w, None, spec, packed=packed)
    elif kind == "dgrad":
        MC.conv_dgrad(dy, w, spec, tuple(x.shape))
    else:
        MC.conv_wgrad(x, dy, tuple(w.shape), spec, want_bias=False)
torch.cuda.synchronize()
