"""liso_region_props on synthetic label maps: python scripts/region_props_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd import _lib as L

dev = torch.device("cuda")
G, K = 512, 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


lib = L.lib()
for name, nblob, size in (("empty", 0, 0), ("30 blobs of 12x12", 30, 12), ("30 blobs of 3x3", 30, 3), ("1 blob of 66x66", 1, 66)):
    lab = torch.zeros((1, G, G), dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(0)
    for k in range(nblob):
        r, c = [int(v) for v in torch.randint(0, G - size, (2,), generator=g)]
        lab[0, r:r + size, c:c + size] = k + 1
    mom = torch.zeros((1, K, 6), dtype=torch.int64, device=dev)
    ws = torch.empty(max(lib.liso_region_props_workspace_bytes(1, G, G, K), 8), dtype=torch.uint8, device=dev)
    props = torch.zeros((1, K, 5), dtype=torch.float64, device=dev)
    t = timeit(lambda: L.check(lib.liso_region_props_ws(L.ptr(lab), 1, G, G, K, L.ptr(mom), L.ptr(props), L.ptr(ws), ws.numel(), L.stream_ptr()), "rp"))
    print(f"{name:22s}: {t:7.1f} us per call (zero fill + moments + props), labelled cells {int((lab > 0).sum())}")
