import os
import sys

import pytest

# SlimTrainer(use_graph=True) requires it (liso_amd/utils/graph_safety.py); must be in the environment before HIP initialises.
# tests/test_gpu_liso_loop.py::test_long_graph_loop_with_recorded_packets runs the loop's graphs WITHOUT it in a child process.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    # the oracle is test infrastructure: make sure its C restatement is compiled (gcc, seconds)
    import oracle

    if not os.path.exists(oracle.ORACLE_SO):
        oracle.build(with_ref=False)
