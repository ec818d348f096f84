"""In-kernel timing of conv_wgrad_rs3_kernel (diagnostic build: make -C liso_amd/csrc STAMPS=1 OUT=../libliso_hip_stamps.so OBJDIR=build_stamps,
copied over libliso_hip.so on the GPU box): per block, s_memtime ticks the MFMA waves spend waiting for the first tile / multiplying / at the
tile barrier / writing the slab, and the loader waves spend converting + storing / issuing loads / at the barrier.
python scripts/wgrad_stamps.py B ci co H [fp32|bf16]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from liso_amd import _lib as L  # noqa: E402
from liso_amd.utils import mfma_conv as MC  # noqa: E402

B, ci, co, H = [int(v) for v in sys.argv[1:5]]
dt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == "fp32") else torch.bfloat16
spec = MC.ConvSpec(3, 3, 1, 1, False)
x = torch.randn(B, ci, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, co, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
for _ in range(5):
    MC.conv_wgrad(x, dy, (co, ci, 3, 3), spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    MC.conv_wgrad(x, dy, (co, ci, 3, 3), spec)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
blocks = 256
buf = (ctypes.c_ulonglong * (blocks * 16))()
fn = L.lib().liso_wgrad_stamps_read
fn.restype = ctypes.c_int
assert fn(buf, blocks) == 0
v = torch.tensor(list(buf), dtype=torch.float64).reshape(blocks, 16)
v = v[v[:, 5] > 0]
names = ["mfma: wait for tile 0", "mfma: multiply (sum)", "mfma: barrier wait (sum)", "mfma: start -> all tiles done", "mfma: slab stores",
         "mfma: whole kernel", "", "", "loader: prologue (load, store, load, barrier)", "loader: wait + convert + store (sum)",
         "loader: issue loads (sum)", "loader: barrier wait (sum)", "tiles of this block"]
print(f"B{B} {ci}->{co} @{H} {dt}: {v.shape[0]} blocks with stamps; wgrad + reduce {us:.1f} us per call (events, eager)")
for i, nm in enumerate(names):
    if nm:
        print(f"  {nm:48s} median {v[:, i].median():10.0f}  min {v[:, i].min():10.0f}  max {v[:, i].max():10.0f} ticks")
