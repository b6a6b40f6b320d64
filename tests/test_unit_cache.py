"""The unit cache of the island kernel (bepuphysics2_amd/csrc/bepu_unit_cache.h) without a GPU: bepuhip_prebuild_unit compiles a unit for a type mask with the build's own
flags, finds it again without a compiler run, and refuses what is not a type set; the units of the BASELINE.json scenes are where build() puts them."""
import ctypes as C
import os
import re
import subprocess
import time

import pytest

from bepuphysics2_amd import build, native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "bepuphysics2_amd", "csrc")


def test_units_are_compiled_with_the_flags_of_the_library():
    text = open(os.path.join(CSRC, "bepu_unit_cache.h")).read()
    flags = re.search(r"kUnitFlags\[\] = \{(.*?)\};", text, flags=re.S).group(1)
    assert re.findall(r'"([^"]+)"', flags) == build.HIP_COMPILE_FLAGS, "a unit compiled at run time must be compiled like the prebuilt ones (bepuphysics2_amd/build.py)"
    sources = re.findall(r'"([^"]+)"', re.search(r"kUnitSources\[\] = \{(.*?)\};", text, flags=re.S).group(1))
    unit = open(os.path.join(CSRC, "bepu_cluster_variant.inc")).read() + "".join(open(os.path.join(CSRC, s)).read() for s in sources)
    included = set(re.findall(r'#include "([^"]+)"', unit))
    assert included <= set(sources), f"the hash of a unit's sources misses {sorted(included - set(sources))}"


def test_prebuild_compiles_once_and_finds_the_object_again(tmp_path, monkeypatch):
    lib = native.load_library()
    monkeypatch.setenv("BEPUHIP_UNIT_CACHE", str(tmp_path))
    path = C.create_string_buffer(1024)
    mask = (1 << 7)  # Contact4 alone
    t0 = time.perf_counter()
    assert lib.bepuhip_prebuild_unit(mask, 1024, 0, path, 1024) == 0, lib.bepuhip_last_error()
    first_s = time.perf_counter() - t0
    built = path.value.decode()
    assert os.path.dirname(built) == str(tmp_path) and re.fullmatch(r"unit_[0-9a-f]{16}_m00000000000080_t1024\.so", os.path.basename(built)), built
    symbols = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    assert " bepu_special_unit" in symbols
    t0 = time.perf_counter()
    assert lib.bepuhip_prebuild_unit(mask, 1024, 0, path, 1024) == 0 and path.value.decode() == built
    assert time.perf_counter() - t0 < min(1.0, first_s), "the second request is a file lookup"
    assert lib.bepuhip_prebuild_unit(0, 1024, 0, path, 1024) == native.BEPUHIP_E_INVALID_ARGUMENT
    assert lib.bepuhip_prebuild_unit(mask, 640, 0, path, 1024) == native.BEPUHIP_E_INVALID_ARGUMENT
    assert lib.bepuhip_prebuild_unit(1 << 12, 1024, 0, path, 1024) == native.BEPUHIP_E_UNSUPPORTED  # 12 is no constraint type id


def test_the_baseline_scenes_units_are_where_build_puts_them():
    units = os.path.join(CSRC, "units")
    if not os.path.isdir(units):
        pytest.skip("build_units() has not run in this tree")
    names = os.listdir(units)
    for mask, budget, split, what in build.BASELINE_UNITS:
        want = f"_m{mask:014x}_t{budget}{'s' if split else ''}.so"
        assert any(n.endswith(want) for n in names), (what, want, names)
