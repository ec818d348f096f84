#!/usr/bin/env python3
"""Library-backed comparison run (NOT a product path): the same trainers with every convolution routed through torch's
convolution (MIOpen on ROCm) instead of the own kernels, eager launches, to put a library number next to the own kernels'.

The package has no backend switch (round 6): this script patches, in ITS OWN process, the two places that keep device tensors on
the own kernels -- `mfma_conv.on_device` (-> False: the modules take their host forward) and `host_ops._HOST_ONLY` (-> False:
that forward accepts device tensors) -- and then runs bench.py's main() with --eager.  Round 2's numbers of this comparison are in
profiles/r02_compare/.

    python scripts/compare_backends.py --steps 10 --warmup 3 --no-cpu-baseline --no-iou3d        (the detector train step, configs[2])

Only the DETECTOR's modules have a host forward to fall back on (rpn.py / center_head.py gate on `mfma_conv.on_device`); SLIM's encoders
and update block call the kernels' entry points directly, so the comparison is the detector workload.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from liso_amd.utils import host_ops  # noqa: E402
from liso_amd.utils import mfma_conv as MC  # noqa: E402

host_ops._HOST_ONLY = False
MC.on_device = lambda x: False
MC.supported = lambda *a, **k: False

import bench  # noqa: E402

if "--workload" not in sys.argv:
    sys.argv += ["--workload", "detector"]
assert sys.argv[sys.argv.index("--workload") + 1] == "detector", "the library-backed comparison covers the detector workload"
if "--eager" not in sys.argv:
    sys.argv.append("--eager")
for flag in ("--no-fp32-leg", "--no-legs"):
    if flag not in sys.argv:
        sys.argv.append(flag)
bench.main()
