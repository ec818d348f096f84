"""Run loop tests with torch.empty() filled with NaN (torch.utils.deterministic.fill_uninitialized_memory under
use_deterministic_algorithms(warn_only)): a kernel that reads memory it never wrote then produces NaN losses / mismatches instead of
depending on whatever the allocator hands out.    python scripts/nan_fill_check.py [pytest -k expression]"""
import os
import sys

import pytest
import torch

torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
k = sys.argv[1] if len(sys.argv) > 1 else "graph_replayed or overlapped"
sys.exit(pytest.main(["tests/test_gpu_liso_loop.py", "-q", "--tb=short", "-x", "-k", k, "-W", "ignore"]))
