import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from liso_amd import _lib as L
from oracle import iou3d as O
g = np.load('tests/golden/iou3d_rand1000_s3.npz')
b = g['boxes_sorted']; n = len(b); cb = (n+63)//64
tb = torch.from_numpy(b).cuda()
lib = L.lib()
ws = torch.zeros(n*cb, dtype=torch.int64, device='cuda')
keep = torch.zeros(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int32, device='cuda')
rc = lib.liso_iou3d_nms_f32(L.ptr(tb), n, 0.1, L.ptr(keep), L.ptr(num), L.ptr(ws), n*cb*8, L.stream_ptr())
torch.cuda.synchronize()
mask = ws.cpu().numpy().view(np.uint64).reshape(n, cb)
iou = g['iou']
ref = np.zeros((n, cb), np.uint64)
for r in range(n):
    for c in np.nonzero(iou[r] > 0.1)[0]:
        if c > r: ref[r, c//64] |= np.uint64(1) << np.uint64(c % 64)
bad = 0
for r in range(n):
    for w in range(r//64, cb):
        if mask[r,w] != ref[r,w]:
            bad += 1
            if bad < 10: print('mask mismatch row', r, 'word', w, hex(mask[r,w]), hex(ref[r,w]))
print('mask mismatches', bad)
k = keep[:int(num.item())].cpu().numpy()
print('num', len(k), 'expected', len(g['keep_010']))
