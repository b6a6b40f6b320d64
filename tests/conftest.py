import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_solver_factory():
    """Creates HipSolver instances; fails loudly if the HIP extension or a GPU is missing (no fallback)."""
    from bepuphysics2_amd.native import HipSolver

    created = []

    def make(**kw):
        s = HipSolver(**kw)
        created.append(s)
        return s

    yield make
    for s in created:
        s.close()
