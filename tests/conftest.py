import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def host_threads():
    """CPU threads this process may really use: min(affinity, cgroup quota) -- NOT os.cpu_count(),
    which on the GPU box reports 256 while the container is capped at 16 cores (oversubscribing
    makes the CPU oracle ~100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


@pytest.fixture(autouse=True, scope="session")
def _torch_threads():
    import torch
    torch.set_num_threads(host_threads())
    yield
