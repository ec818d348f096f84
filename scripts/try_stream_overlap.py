"""do kernels of two HIP streams run concurrently here?  the same small convolution (192 blocks) on 1 stream vs on 2 / 3"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.utils import mfma_conv as MC

dev = torch.device("cuda")
conv = torch.nn.Conv2d(128, 128, 3, padding=1).to(dev)
for p in conv.parameters():
    p.requires_grad_(False)
xs = [torch.randn(1, 128, 64, 64, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
R = 200


def run(n_streams, graph=False):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for r in range(R):
        for s in range(n_streams):
            with torch.cuda.stream(streams[s]), torch.no_grad():
                MC.conv2d(conv, xs[s])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


def run_graphs(n_streams):
    gs = []
    for s in range(n_streams):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(streams[s]), torch.no_grad():
            MC.conv2d(conv, xs[s])
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=streams[s]), torch.no_grad():
            for _ in range(R):
                MC.conv2d(conv, xs[s])
        gs.append(g)
    out = {}
    for n in range(1, n_streams + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for s in range(n):
            with torch.cuda.stream(streams[s]):
                gs[s].replay()
        torch.cuda.synchronize()
        out[n] = (time.perf_counter() - t) * 1e3
    return out


run(1)
for n in (1, 2, 3):
    print(f"eager, {n} stream(s): {run(n):.2f} ms for {R} convs per stream")
print("graphs of", R, "convs:", {k: round(v, 2) for k, v in run_graphs(3).items()})
