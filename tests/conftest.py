import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads_from_quota():
    """PyTorch sizes its CPU pools from the affinity mask (256 CPUs on the GPU boxes) although the cgroup quota allows 16: the
    CPU oracle then runs oversubscribed and several times slower.  Size the pools from the quota, as bench.py does."""
    import torch
    from mrn_amd.tools.utils import host_cpu_budget
    torch.set_num_threads(max(1, min(host_cpu_budget(), 32)))
    yield
