"""Can a 2-MFMA-per-product arithmetic be certified at 1e-3?  (round-5 VERDICT item 1: "fp16 hi + fp16 residual, or weights in one fp16
plane" instead of F32X3's three bf16 products.)  Measured BEFORE building the kernel mode: a 2-term product drops one operand's
residual, i.e. it computes with that operand ROUNDED to the 11 significand bits of its fp16 high plane:
    (a_h + a_l) . w_h            == F32X3-or-better arithmetic on weights rounded to fp16            -> rows "weights -> N bits"
    a_h . (w_h + w_l)            == ... on the convolutions' input activations rounded to fp16       -> row  "inputs -> fp16"
Everything else runs F32X3 on fp32 tensors (whose own error, 1e-4 at this size, is the floor of every row).  Truth = exact fp32 MFMA on
the unrounded weights.  120k-point cloud, 512^2 BEV, B = 1, train-mode BatchNorm, random-init weights, as mixed_precision_budget.py.
Columns: worst raw-logit error of every head relative to the head's largest logit (the measure of tests/test_gpu_parity_full_size.py).
python scripts/two_term_budget.py"""
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
model = tr.net.model
convs = [(n, m) for n, m in list(model.rpn.named_modules()) + list(model.center_head.named_modules())
         if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
backbone = [(n, m) for n, m in convs if n.startswith("blocks.")]


def round_bits(w, bits):
    """round to `bits` significand bits (11 = fp16's, 8 = bf16's, 16 = a bf16 hi/lo pair's, 24 = fp32)"""
    if bits == 11:
        return w.half().float()
    m, e = torch.frexp(w)
    return torch.ldexp(torch.round(m * 2.0 ** bits) / 2.0 ** bits, e)


orig = MC.fused_conv
round_inputs = [False]


def hooked(x_raw, fold, conv, *a, **k):
    if round_inputs[0] and x_raw.dtype == torch.float32:
        x_raw = x_raw.half().float()
    return orig(x_raw, fold, conv, *a, **k)


def logits(mode, rounded=(), bits=11, inputs=False):
    saved = [(m, m.weight.detach().clone()) for _, m in rounded]
    with torch.no_grad():
        for _, m in rounded:
            m.weight.copy_(round_bits(m.weight, bits))
    round_inputs[0] = inputs
    prev = MC.set_fp32_mode(mode)
    MC.fused_conv = hooked
    try:
        with torch.no_grad():
            _, _, raw, _ = tr.net(None, pcls, None, decode=False)
    finally:
        MC.fused_conv = orig
        MC.set_fp32_mode(prev)
        round_inputs[0] = False
        with torch.no_grad():
            for m, w in saved:
                m.weight.copy_(w)
    return {h: raw[h].detach().double() for h in HEADS}


truth = logits("exact")


def row(name, got):
    errs = [float((got[h] - truth[h]).abs().max() / truth[h].abs().max()) for h in HEADS]
    print(f"{name:58s} " + "  ".join(f"{h} {e:8.2e}" for h, e in zip(HEADS, errs)) + f"   worst {max(errs):8.2e} {'<= 1e-3' if max(errs) <= 1e-3 else '> 1e-3'}", flush=True)
    return max(errs)


print("worst raw-logit error / largest logit of the head, against exact fp32 MFMA (120k points, 512^2, B = 1, train-mode BatchNorm)")
row("F32X3 everywhere (3 products, 16 bits per operand)", logits("x3"))
for bits in (11, 12, 13, 14, 15, 16):
    tag = " = one fp16 plane: the 2-term product" if bits == 11 else ""
    row(f"weights of every convolution -> {bits} bits{tag}", logits("x3", convs, bits))
row("inputs of every convolution -> fp16 (a_h . (w_h + w_l))", logits("x3", inputs=True))
row("weights -> fp16, exact fp32 MFMA (no F32X3 noise under it)", logits("exact", convs, 11))
single = []
for name, m in backbone:
    single.append((row(f"weights -> fp16: ONLY {name} {m.in_channels}->{m.out_channels}", logits("x3", [(name, m)], 11)), name))
best = min(single)
print(f"best single backbone layer with fp16 weights: {best[1]} -> {best[0]:.2e}")
