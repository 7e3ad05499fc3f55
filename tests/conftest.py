import os
import sys

import pytest

# torch's intra-op pool defaults to every hardware thread of the node; under a cgroup CPU
# quota their spinning gets the whole test process throttled (see bench.py)
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# ... and the oracle's OpenMP pool: confined to as many CPUs as the quota pays for, the test
# process cannot overrun it.  (The GPU suite runs in ~40 s; it took 199 s while importing
# bench.py pinned the whole session to four CPUs -- bench.py pins only as a program now.)
try:
    from msmdfusion_amd.hostcpu import within_quota
    if hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, within_quota(os.sched_getaffinity(0)))
except (ImportError, OSError):
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
