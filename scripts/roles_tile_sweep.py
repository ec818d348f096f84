"""conv_roles_kernel: tile shape sweep on the layers of the RAFT update block at inference (1, 2, 4 sweep pairs per replay).
One child process per (MI, NJ) -- the planner's switches are read once --, hipGraph of 20 launches, median of 5 replays.
    python scripts/roles_tile_sweep.py            # the table
(Round 5 ran this sweep with a third axis, the channel slabs of a tile split over 2-4 compute units with a ticketed hand-over through
memory: profiles/r05_roles_split_sweep.txt, DESIGN.md section 10 -- removed, the hand-over cost more than the idle CUs.)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(4, 304, 192, 64, 64), (4, 304, 96, 64, 64), (4, 160, 80, 64, 64), (4, 128, 64, 64, 64), (4, 256, 6, 64, 64), (4, 96, 256, 64, 64),
          (2, 304, 192, 64, 64), (2, 304, 96, 64, 64), (2, 160, 80, 64, 64), (1, 304, 192, 64, 64), (1, 304, 96, 64, 64), (1, 256, 6, 64, 64)]

CHILD = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from liso_amd import _lib as L
from liso_amd.utils import mfma_conv as MC
shapes = %r
spec = MC.ConvSpec(3, 3, 1, 1, False)
for (B, Ci, Co, H, W) in shapes:
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.02
    packed = MC.pack_weights(w, spec, False, MC._mode(x.dtype))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            MC.conv_forward(x, w, None, spec, packed=packed)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            MC.conv_forward(x, w, None, spec, packed=packed)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    xv, xps = MC.as_nhwc(x, 4)
    d = MC.gather_desc(spec, B, H, W, Ci, xps, H, W, Co, Co, 0, L.CONV_F32X3, True, False, False)
    info = (ctypes.c_int * 8)()
    L.lib().liso_conv_plan_info(ctypes.byref(d), info)
    print("RES", B, Ci, Co, H, W, "plan", info[1], info[2], "items", info[4], "us %%.1f" %% sorted(ts)[2], flush=True)
"""


def main():
    combos = [(0, 0)] + [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2)]
    rows = {}
    for mi, nj in combos:
        env = dict(os.environ)
        if mi:
            env.update(LISO_ROLES_MI=str(mi), LISO_ROLES_NJ=str(nj))
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, SHAPES)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print("FAILED", mi, nj, r.stderr[-500:])
            continue
        for line in r.stdout.splitlines():
            if line.startswith("RES"):
                f = line.split()
                shape = tuple(int(v) for v in f[1:6])
                rows.setdefault(shape, []).append(((mi, nj), tuple(int(v) for v in f[7:9]), int(f[10]), float(f[12])))
    for shape, lst in rows.items():
        print("x%s" % (shape,))
        for forced, plan, items, us in lst:
            print("   forced %s -> plan (mi %d, nj %d) tiles %4d : %6.1f us" % ("auto  " if forced[0] == 0 else forced, *plan, items, us))


if __name__ == "__main__":
    main()
