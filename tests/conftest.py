import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Device-group tests run several members on this box's ONE GPU beside whatever contexts other tests keep alive; members wait for each other inside their kernels and need
# a hardware queue each (include/bepuhip.h, device groups (4)). Read by the HIP runtime when it starts: set before anything touches it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def hip_solver_factory():
    """Creates HipSolver instances; fails loudly if the HIP extension or a GPU is missing (no fallback). Per test (round 6): a context holds a stream, streams share the
    device's hardware queues, and a session's worth of idle contexts is exactly what a device-group test on one GPU cannot have beside it (include/bepuhip.h, device groups (4))."""
    from bepuphysics2_amd.native import HipSolver

    created = []

    def make(**kw):
        s = HipSolver(**kw)
        created.append(s)
        return s

    yield make
    for s in created:
        s.close()
