"""Fill the caching allocator with poisoned blocks, then run the detector step and list gradient errors per parameter."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "nan"
if mode != "clean":
    blocks = []
    for sz in [2 ** k for k in range(9, 29)] + [3 * 2 ** k for k in range(9, 27)]:
        for _ in range(3):
            if mode == "nan":
                blocks.append(torch.full((sz // 4,), float("nan"), device="cuda"))
            else:
                blocks.append(torch.full((sz // 4,), 12345678, dtype=torch.int32, device="cuda"))
    del blocks
import test_gpu_detector as T
from oracle.train_step import detector_forward_loss, prepare_state
tr, pcls, targets = T._setup(128, 100.0, 2, 20000)
sd0 = tr.net.state_dict()
tr.model.train()
total, losses, _ = tr.loss(pcls, targets)
total.backward()
sd64 = prepare_state(sd0, torch.float64)
ref64, _, _ = detector_forward_loss(sd64, [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}, 128, 100.0, dtype=torch.float64)
ref64.backward()
print("loss", float(total), float(ref64))
for k, p in tr.net.named_parameters():
    if p.grad is not None and sd64[k].grad is not None and float(sd64[k].grad.abs().max()) >= 1e-6:
        e = T._rel(p.grad, sd64[k].grad)
        if e > 1e-3:
            print("%-60s %.3e" % (k, e))
