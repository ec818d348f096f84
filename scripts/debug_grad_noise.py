"""GPU fp32 detector-step gradients vs the fp64 CPU oracle, next to the fp32 CPU oracle's own error (conditioning)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_gpu_detector as T
if os.environ.get("DET"): torch.backends.cudnn.deterministic = True; print("deterministic convs")
if os.environ.get("BENCHMARK"): torch.backends.cudnn.benchmark = True
from oracle.train_step import detector_forward_loss, prepare_state
print("threads", torch.get_num_threads(), "mkldnn", torch.backends.mkldnn.is_available(), os.environ.get("MIOPEN_DEBUG_CONV_WINOGRAD"))
import conftest, contextlib
ctx = contextlib.nullcontext()
if os.environ.get("EXACT"):
    gen = conftest.exact_convs.__wrapped__() if hasattr(conftest.exact_convs, "__wrapped__") else None
    next(gen); print("exact convs")
if os.environ.get("TORCHBN"):
    import liso_amd.networks.centerpoint.fused_bn as fb
    fb._supported = lambda c, d: False; print("torch BN")
tr, pcls, targets = T._setup(128, 100.0, 2, 20000)
sd0 = tr.net.state_dict()
tr.model.train()
total, losses, _ = tr.loss(pcls, targets)
total.backward()
cp, tc = [p.cpu() for p in pcls], {k: v.cpu() for k, v in targets.items()}
sd64 = prepare_state(sd0, torch.float64)
ref64, _, _ = detector_forward_loss(sd64, cp, tc, 128, 100.0, dtype=torch.float64); ref64.backward()
sd32 = prepare_state(sd0, torch.float32)
ref32, _, _ = detector_forward_loss(sd32, cp, tc, 128, 100.0); ref32.backward()
worst_g = worst_c = 0
for k, p in tr.net.named_parameters():
    if p.grad is not None and sd64[k].grad is not None and float(sd64[k].grad.abs().max()) >= 1e-6:
        eg, ec = T._rel(p.grad, sd64[k].grad), T._rel(sd32[k].grad, sd64[k].grad)
        worst_g, worst_c = max(worst_g, eg), max(worst_c, ec)
        if 'center_head' in k:
            print("%-58s gpu %.2e  cpu32 %.2e" % (k, eg, ec))
print("loss gpu %.6f cpu32 %.6f cpu64 %.6f | worst gpu %.2e worst cpu32 %.2e" % (float(total), float(ref32), float(ref64), worst_g, worst_c))
