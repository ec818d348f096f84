"""Per-layer bf16 error budget of the detector at BASELINE size (round-4 VERDICT item 3: "a configuration <= 1e-3 on logits at <= 6.0 ms
per step, or a table proving none exists").  120k-point cloud, 512^2 BEV, B = 1, train-mode BatchNorm, random-init weights.
Truth = exact fp32 MFMA.  Rows: F32X3 everywhere; ONE backbone convolution in bf16 (its input rounded to bf16, bf16 MFMAs, bf16 output,
everything else F32X3 on fp32 tensors); the plan the VERDICT sketches (stride-2 layers + deblocks + heads F32X3, the stride-1 backbone
layers bf16); all bf16.  Columns: worst raw-logit error of every head relative to the head's largest logit.
python scripts/mixed_precision_budget.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import detector_batch
from liso_amd.trainer import DetectorTrainer
from liso_amd.utils import mfma_conv as MC
from liso_amd.utils.config import default_cfg

HEADS = ("pos", "dims", "rot", "probs")
dev = torch.device("cuda:0")
torch.manual_seed(11)
tr = DetectorTrainer(default_cfg(grid=512, bev_range_m=100.0), dev, compute_dtype=torch.float32, total_steps=20)
pcls, _ = detector_batch(16, 1, dev, n_points=120000, grid=512, bev_range_m=100.0)
tr.model.train()
rpn = tr.net.model.rpn
layers = []  # (name, conv module) of the backbone blocks
for i, block in enumerate(rpn.blocks):
    mods = list(block)
    layers.append((f"block{i}.conv0 {mods[1].in_channels}->{mods[1].out_channels} stride {mods[1].stride[0]}", mods[1]))
    for j in range(4, len(mods), 3):
        layers.append((f"block{i}.conv{(j - 1) // 3} {mods[j].in_channels}->{mods[j].out_channels}", mods[j]))

orig = MC.fused_conv
in_bf16 = set()


def hooked(x_raw, fold, conv, *a, **k):
    if id(conv) in in_bf16 and k.get("out") is None:
        y, f = orig(x_raw.to(torch.bfloat16), fold, conv, *a, **k)  # (the pending BatchNorm fold holds fp32 vectors: applied in the prologue)
        return y.float(), f
    return orig(x_raw, fold, conv, *a, **k)


def logits(mode, which):
    in_bf16.clear()
    in_bf16.update(id(c) for c in which)
    prev = MC.set_fp32_mode(mode)
    MC.fused_conv = hooked
    try:
        with torch.no_grad():
            _, _, raw, _ = tr.net(None, pcls, None, decode=False)
    finally:
        MC.fused_conv = orig
        MC.set_fp32_mode(prev)
    return {h: raw[h].detach().double() for h in HEADS}


truth = logits("exact", [])


def row(name, got):
    errs = [float((got[h] - truth[h]).abs().max() / truth[h].abs().max()) for h in HEADS]
    print(f"{name:44s} " + "  ".join(f"{h} {e:8.2e}" for h, e in zip(HEADS, errs)) + f"   worst {max(errs):8.2e} {'<= 1e-3' if max(errs) <= 1e-3 else '> 1e-3'}")
    return max(errs)


print("worst raw-logit error / largest logit of the head, against exact fp32 MFMA (120k points, 512^2, B = 1, train-mode BatchNorm)")
row("F32X3 everywhere", logits("x3", []))
single = []
for name, conv in layers:
    single.append((row("bf16: " + name, logits("x3", [conv])), name))
stride1 = [c for n, c in layers if "stride 2" not in n and "conv0" not in n]
row("bf16: every stride-1 backbone layer", logits("x3", stride1))
row("bf16: every backbone layer", logits("x3", [c for _, c in layers]))
best = min(single)
print(f"best single layer in bf16: {best[1]} -> {best[0]:.2e}: " + ("a mixed plan within 1e-3 may exist" if best[0] <= 1e-3 else
      "NO plan that runs even one backbone layer in bf16 stays within 1e-3 on the logits"))
