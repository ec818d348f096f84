"""CPU: the host-side planning arithmetic of the convolution launches (liso_amd/csrc/conv_plan.h) built with
-fsanitize=address,undefined and run over 220 000 pseudo-random + degenerate descriptors (tests/san/conv_plan_driver.cpp): every
accepted plan satisfies the invariants the kernels rely on (LDS within 160 KB, tiles cover the output, tap indices inside the packed
weights), refused descriptors are refused without touching memory.  GPU AddressSanitizer is not available on this pool: the device
side is covered by the guard-band runs of tests/test_gpu_canaries.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_conv_plan_arithmetic_under_asan_ubsan(tmp_path):
    exe = tmp_path / "conv_plan_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wno-unused-function",
           os.path.join(ROOT, "tests", "san", "conv_plan_driver.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for k in list(env):  # the experiment switches of the planner must not leak in from the caller's environment
        if k.startswith("LISO_"):
            del env[k]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("OK "), (r.stdout[-2000:], r.stderr[-3000:])
