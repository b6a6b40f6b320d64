"""bench.py's extra legs run in child processes (bench.leg_in_child): a leg that cannot produce its report — here: no GPU in the test container — comes back as an error
entry, it does not raise and it does not take the caller down. The headline's JSON line depends on that (profiles/r05_s30_bench_fault.txt)."""
import argparse
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_a_leg_that_dies_is_reported_not_raised():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU: with one the leg simply runs")
    import bench
    args = argparse.Namespace(ragdolls=50, no_traffic=True)
    report = bench.leg_in_child("lattice", args, 0, timeout=120.0)
    assert isinstance(report, dict) and "error" in report
    assert "lattice" in report["error"]


def test_the_leg_names_are_the_ones_the_main_process_asks_for():
    import bench
    source = open(os.path.join(REPO, "bench.py")).read()
    for name in bench.EXTRA_LEGS:
        assert f'extra("{name}")' in source, name
