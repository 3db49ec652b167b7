import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The CPU oracle is the slow half of the GPU suite.  A GPU box hands a one-GPU job 16 CPUs of a 256-CPU host, and torch sizes its
    intra-op pool by what the host reports: on a busy host the over-subscribed pool made single oracle tests take minutes (one run was
    killed by the harness's 7-minute silence rule).  Pin the pool to the CPUs this process may actually use, at most 16."""
    try:
        import torch

        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(n, 16)))
    except Exception:      # noqa: BLE001 - a missing torch fails the tests that need it, not the collection
        pass


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
