"""one convolution layer, forward only, graph-timed: python scripts/one_layer.py B ci co H k [fp32|bf16]"""
import sys, torch
from liso_amd.utils import mfma_conv as MC
B, ci, co, H, k = [int(v) for v in sys.argv[1:6]]
dt = torch.float32 if (len(sys.argv) > 6 and sys.argv[6] == "fp32") else torch.bfloat16
spec = MC.ConvSpec(k, k, 1, k // 2, False)
x = torch.randn(B, ci, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, k, k, device="cuda") * 0.05
packed = MC.pack_weights(w, spec, False, MC._mode(dt))
fn = lambda: MC.conv_forward(x, w, None, spec, packed=packed)
for _ in range(3): fn()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    fn()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20): fn()
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 20 * 1e3)
fl = 2.0 * B * H * H * co * ci * k * k
print(f"B{B} {ci}->{co} @{H} k{k} {sys.argv[6] if len(sys.argv) > 6 else 'bf16'}: {best:7.1f} us  {fl / best / 1e6:7.1f} TF")
