"""Per-layer table of the convolutions of ONE SLIM inference replay of the LISO loop (`--ib` pairs per launch, F32X3): every
`conv_forward` call of an eager pass is recorded (shape, prologue, statistics), then each distinct layer is timed alone (20 launches in
one hipGraph) -> us, algorithmic TFLOP/s, launches per replay, share of the replay's convolution time.  `--det` does the same for the
detector forward at `--batch` pairs in the current arithmetic (`--dtype bf16|fp32`)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils import mfma_conv as MC
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

arg = lambda k, d: int(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d  # noqa: E731
IB = arg("--ib", 4)
dev = torch.device("cuda")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=400, use_graph=False, overlap=False, infer_batch=IB)
pairs = [slim_pair(2 + 100 * i, dev) for i in range(IB)]

calls = collections.OrderedDict()
orig = MC.conv_forward


def hook(x, weight, bias, spec, in_scale=None, in_shift=None, in_relu=False, out_relu=False, out_dtype=None, want_stats=False,
         stats_shift=None, packed=None, affine_batch_stride=0, occupancy=None, out=None):
    key = (tuple(x.shape), str(x.dtype), tuple(weight.shape), spec.kh, spec.kw, spec.stride, spec.padding, spec.transposed,
           in_scale is not None, bool(in_relu), bool(out_relu), bool(want_stats), int(affine_batch_stride), occupancy is not None,
           out is not None)
    calls[key] = calls.get(key, 0) + 1
    return orig(x, weight, bias, spec, in_scale, in_shift, in_relu, out_relu, out_dtype, want_stats, stats_shift, packed,
                affine_batch_stride, occupancy, out)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e-3)
    return best


MC.conv_forward = hook
with torch.no_grad():
    s0 = tr._stack_samples([p[0] for p in pairs]); s1 = tr._stack_samples([p[1] for p in pairs])
    tr._infer_flow(s0, s1)
torch.cuda.synchronize()
MC.conv_forward = orig

rows = []
for key, n in calls.items():
    (xs, xdt, ws, kh, kw, stride, padding, transposed, pro, in_relu, out_relu, stats, abs_, occ, has_out) = key
    dt = torch.float32 if "float32" in xdt else torch.bfloat16
    spec = MC.ConvSpec(kh, kw, stride, padding, transposed)
    B, ci, H, W = xs
    x = torch.randn(B, ci, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(ws, device=dev) * 0.05
    co = ws[1] if transposed else ws[0]
    packed = MC.pack_weights(w, spec, False, MC._mode(dt))
    if pro:
        nv = ci * (B if abs_ else 1)
        sc, sh = torch.rand(nv, device=dev) + 0.5, torch.randn(nv, device=dev)
    else:
        sc = sh = None
    ho, wo = spec.out_hw(H, W)
    fl = 2.0 * B * ho * wo * co * ci * kh * kw / (stride * stride if transposed else 1)
    t = timeit(lambda: orig(x, w, None, spec, sc, sh, in_relu, out_relu, None, stats, None, packed, abs_ if pro else 0, None, None))
    rows.append((key, n, t, fl))
tot = sum(n * t for _, n, t, _ in rows)
print(f"convolution launches per replay of {IB} pairs: {sum(n for _, n, _, _ in rows)}; sum of isolated times {tot * 1e3:.3f} ms; "
      f"sum of flops {sum(n * f for _, n, _, f in rows) / 1e9:.1f} GFLOP -> {sum(n * f for _, n, _, f in rows) / tot / 1e12:.1f} TFLOP/s")
for key, n, t, fl in sorted(rows, key=lambda r: -r[1] * r[2]):
    (xs, xdt, ws, kh, kw, stride, padding, transposed, pro, in_relu, out_relu, stats, abs_, occ, has_out) = key
    print(f"x{list(xs)} {xdt[6:]:8s} w{list(ws)} s{stride} {'pro ' if pro else ''}{'stats ' if stats else ''}{'occ ' if occ else ''}"
          f"{'out ' if has_out else ''}: {n:3d} x {t * 1e6:7.1f} us = {n * t * 1e3:6.3f} ms ({100 * n * t / tot:4.1f} %)  {fl / t / 1e12:6.1f} TFLOP/s")
