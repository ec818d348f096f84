"""The loop and detector graphs under the runtime's DEFAULT graph mode (recorded packets), which is what bench.py runs in.

tests/conftest.py sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for the whole session because the SLIM *training* graph needs node-by-node
replays (liso_amd/utils/graph_safety.py) and the variable must be in the environment before HIP initialises.  The loop's three graphs
and the detector's hold no memset node and do not need it: this test re-runs their test files in a child process with the variable
set to the runtime's default, so that every loop / detector graph test also passes in the mode the benchmark uses."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1500)
def test_loop_and_detector_test_files_pass_with_recorded_graph_packets():
    if os.environ.get("LISO_IN_DEFAULT_MODE_CHILD"):
        pytest.skip("child session")
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1", LISO_IN_DEFAULT_MODE_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_liso_loop.py", "tests/test_gpu_detector.py", "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1400)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
