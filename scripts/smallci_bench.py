import torch, sys, os
sys.path.insert(0, os.getcwd())
from liso_amd.utils import mfma_conv as MC
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (2, 12):
    x = torch.randn(B, 4, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, 64, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    xn, dyn = x.contiguous(), dy.contiguous()
    w = torch.randn(64, 4, 7, 7, device="cuda")
    spec = MC.ConvSpec(7, 7, 1, 3, False)
    t_own = timeit(lambda: MC.conv_wgrad(x, dy, (64, 4, 7, 7), spec, want_bias=True))
    t_own_nchw = timeit(lambda: MC.conv_wgrad(xn, dyn, (64, 4, 7, 7), spec, want_bias=True))
    t_lib = timeit(lambda: torch.ops.aten.convolution_backward(dyn, xn, w, [64], [1, 1], [3, 3], [1, 1], False, [0, 0], 1, [False, True, True]))
    t_lib_cl = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, [64], [1, 1], [3, 3], [1, 1], False, [0, 0], 1, [False, True, True]))
    print(f"B={B}: own (NHWC in) {t_own:.1f} us, own (NCHW in) {t_own_nchw:.1f} us, library NCHW {t_lib:.1f} us, library channels-last {t_lib_cl:.1f} us")
