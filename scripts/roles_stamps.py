"""In-kernel timing of conv_roles_kernel (a diagnostic build: make -C liso_amd/csrc STAMPS=1): per block, shader cycles the MFMA waves
spend multiplying / at the slab barrier and the loader waves spend issuing loads / converting + storing / at the barrier.
python scripts/roles_stamps.py B ci co H [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.utils import mfma_conv as MC
B, ci, co, H = [int(v) for v in sys.argv[1:5]]
dt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == "fp32") else torch.bfloat16
spec = MC.ConvSpec(3, 3, 1, 1, False)
x = torch.randn(B, ci, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
packed = MC.pack_weights(w, spec, False, MC._mode(dt))
for _ in range(5):
    y, st = MC.conv_forward(x, w, None, spec, packed=packed, want_stats=True)
torch.cuda.synchronize()
n = st.numel() * 4 // 8
v = st.reshape(-1).view(torch.int64)[: (n // 16) * 16].reshape(-1, 16).cpu().double()
blocks = min(v.shape[0], 4096)
v = v[:blocks]
v = v[v[:, 3] > 0]
names = ["mfma: wait for buffer 0", "mfma: multiply (sum)", "mfma: barrier wait (sum)", "mfma: start -> loop end", "epilogue", "", "", "",
         "loader: prologue (2 issues + store + barrier)", "loader: issue (sum)", "loader: wait + convert + store (sum)", "loader: barrier wait (sum)"]
print(f"B{B} {ci}->{co} @{H} {dt}: {v.shape[0]} blocks with stamps")
for i, nm in enumerate(names):
    if nm:
        print(f"  {nm:48s} median {v[:, i].median():10.0f}  min {v[:, i].min():10.0f}  max {v[:, i].max():10.0f} cycles")
