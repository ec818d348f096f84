"""Does a hipGraph replay signal an EXTERNAL event from the middle of the graph (torch.cuda.Event(external=True) recorded during
capture -> event-record node), so that another stream can start work (a gradient-bucket all-reduce) before the replay has finished?
Prints the order of completion and checks the data dependency."""
import sys
import time

import torch

dev = torch.device("cuda")
n = 1 << 26
a = torch.zeros(n, device=dev)
b = torch.zeros(n, device=dev)
out = torch.zeros(n, device=dev)
ev = torch.cuda.Event(external=True)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()


def body():
    a.fill_(1.0)
    a.mul_(3.0)  # "first part of backward": a == 3
    ev.record()  # (on the capturing stream)
    for _ in range(40):  # "rest of backward": ~40 passes over 256 MB
        b.add_(1.0)


cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=cap):
        body()
except Exception as e:  # noqa: BLE001
    print("capture with an external event FAILED:", type(e).__name__, str(e)[:300])
    sys.exit(1)
ok = True
for it in range(3):
    a.zero_(), out.zero_()
    torch.cuda.synchronize()
    e_side, e_main = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    g.replay()
    e_main.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        out.copy_(a)
        e_side.record()
    torch.cuda.synchronize()
    good = bool((out == 3.0).all())
    print(f"replay {it}: side copy saw a == 3: {good}; side done at {t0.elapsed_time(e_side):.3f} ms, graph done at {t0.elapsed_time(e_main):.3f} ms")
    ok = ok and good and t0.elapsed_time(e_side) < t0.elapsed_time(e_main)
print("EXTERNAL_EVENT_OK" if ok else "EXTERNAL_EVENT_NOT_USABLE")
