"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms (plain C via ctypes, numpy, or fp32/fp64 torch-CPU),
each citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; liso_amd/ never does.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libiou3d_ref.so")


def build(with_ref=True):
    """Compile liboracle.so (gcc) and, when /root/reference exists, oracle/_ref (unmodified reference TU)."""
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if with_ref:
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        build(with_ref=False)
    return ctypes.CDLL(ORACLE_SO)


def load_ref():
    """The compiled reference TU, or None when it was never built (e.g. /root/reference absent and no prebuilt)."""
    if not os.path.exists(REF_SO):
        return None
    import torch  # noqa: F401  libtorch must be mapped before the reference TU's shim
    return ctypes.CDLL(REF_SO)
