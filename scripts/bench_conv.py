"""micro-benchmark of the own convolution kernels on the detector's layer shapes: forward / dgrad / wgrad, TFLOP/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.utils import mfma_conv as MC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
if len(sys.argv) >= 3 and sys.argv[2] == "exact":  # fp32 tensors on the native fp32 MFMA instead of bf16 hi/lo pairs
    MC.set_fp32_mode("exact")
DET32 = len(sys.argv) >= 4 and sys.argv[3] == "det"  # fp32 / exact on the DETECTOR's layer shapes
LAYERS = [  # name, Ci, Co, H (input), k, s, p, transposed
    ("b0.s2 64->64 512", 64, 64, 512, 3, 2, 1, False), ("b0 64->64 256", 64, 64, 256, 3, 1, 1, False),
    ("b1.s2 64->128 256", 64, 128, 256, 3, 2, 1, False), ("b1 128->128 128", 128, 128, 128, 3, 1, 1, False),
    ("b2.s2 128->256 128", 128, 256, 128, 3, 2, 1, False), ("b2 256->256 64", 256, 256, 64, 3, 1, 1, False),
    ("de0 k2s2 64->128 256", 64, 128, 256, 2, 2, 0, False), ("de1 1x1 128->128", 128, 128, 128, 1, 1, 0, False),
    ("de2 T 256->128 64", 256, 128, 64, 2, 2, 0, True), ("head 384->64 128", 384, 64, 128, 3, 1, 1, False),
    ("heads 64->256 128", 64, 256, 128, 3, 1, 1, False), ("out 64->3 128", 64, 3, 128, 3, 1, 1, False),
]
if dt == torch.float32 and not DET32:
    LAYERS = [("enc 7x7s2 64->32 512", 64, 32, 512, 7, 2, 3, False), ("enc 32->32 256", 32, 32, 256, 3, 1, 1, False),
              ("enc 64->64 128", 64, 64, 128, 3, 1, 1, False), ("enc 96->96 64", 96, 96, 64, 3, 1, 1, False),
              ("corr 1x1 196->96 64", 196, 96, 64, 1, 1, 0, False), ("gru zr 400->192 64", 400, 192, 64, 3, 1, 1, False),
              ("gru q 400->96 64", 400, 96, 64, 3, 1, 1, False), ("head 96->128 64", 96, 128, 64, 3, 1, 1, False),
              ("conv 160->80 64", 160, 80, 64, 3, 1, 1, False), ("flow2 64->32 64", 64, 32, 64, 3, 1, 1, False)]


def timeit(fn, n=20):
    """GPU time per call: n calls captured into ONE hipGraph and replayed (the python wrapper costs ~20 us per call -- more than most
    of these kernels at batch 1 -- so eager loops measure the host)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e-3)
    return best


for name, ci, co, H, k, s, p, tr in LAYERS:
    spec = MC.ConvSpec(k, k, s, p, tr)
    x = torch.randn(B, ci, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn((ci, co, k, k) if tr else (co, ci, k, k), device="cuda") * 0.05
    ho, wo = spec.out_hw(H, H)
    dy = torch.randn(B, co, ho, wo, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    mode = MC._mode(dt)
    packed = MC.pack_weights(w, spec, False, mode)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda")
    fl = 2.0 * B * ho * wo * co * ci * (k * k if not tr else 1)
    t_f = timeit(lambda: MC.conv_forward(x, w, None, spec, packed=packed))
    t_fp = timeit(lambda: MC.conv_forward(x, w, None, spec, sc, sh, in_relu=True, want_stats=True, packed=packed))
    t_d = timeit(lambda: MC.conv_dgrad(dy, w, spec, tuple(x.shape)))
    t_w = timeit(lambda: MC.conv_wgrad(x, dy, tuple(w.shape), spec, want_bias=False))
    wt = w.to(dt)
    conv = (lambda: torch.nn.functional.conv_transpose2d(x, wt, stride=s, padding=p)) if tr else \
        (lambda: torch.nn.functional.conv2d(x, wt, stride=s, padding=p))
    t_m = timeit(conv)
    # the library's (MIOpen's) weight gradient and data gradient on the same tensors
    def lib_bwd(mask):
        return torch.ops.aten.convolution_backward(dy, x, wt, None, [s, s], [p, p], [1, 1], tr, [0, 0], 1, mask)
    try:
        t_mw, t_md = timeit(lambda: lib_bwd([False, True, False])), timeit(lambda: lib_bwd([True, False, False]))
    except Exception as e:  # (a geometry the library refuses)
        t_mw = t_md = float("nan")
    print(f"{name:24s} fwd {t_f*1e6:7.1f} us {fl/t_f/1e12:6.1f} TF | +bn-prologue+stats {t_fp*1e6:7.1f} us | dgrad {t_d*1e6:7.1f} us "
          f"{fl/t_d/1e12:6.1f} TF | wgrad {t_w*1e6:7.1f} us {fl/t_w/1e12:6.1f} TF | torch fwd {t_m*1e6:7.1f} us {fl/t_m/1e12:6.1f} TF"
          f" dgrad {t_md*1e6:7.1f} us wgrad {t_mw*1e6:7.1f} us")
