"""CPU tests of the C-ABI library: it loads, exports every symbol include/bepuhip.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from bepuphysics2_amd import native
from bepuphysics2_amd.scene import TYPE_TABLE

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(REPO, "include", "bepuhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bepuhip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(native.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(native.EXPORTED_SYMBOLS) == declared


def test_type_info_table():
    for tid, (nb, pf, imf, _) in TYPE_TABLE.items():
        assert native.type_info(tid) == (nb, pf, imf)
    with pytest.raises(native.UnsupportedError):
        native.type_info(11)  # ids 11-14 and 18-21 are unused by the reference (DefaultTypes.cs)


def test_create_fails_loudly_without_gpu_or_bad_config():
    import torch
    lib = native.load_library()
    ctx = C.c_void_p()
    bad = native.Config(0, 7, 0)
    assert lib.bepuhip_create(C.byref(bad), C.byref(ctx)) == native.BEPUHIP_E_INVALID_ARGUMENT
    if not torch.cuda.is_available():
        with pytest.raises(native.BepuHipError) as e:
            native.HipSolver()
        assert e.value.code == native.BEPUHIP_E_DEVICE and b"no CPU fallback" in lib.bepuhip_last_error()


def test_every_entry_point_is_mapped_to_the_reference_in_the_integration_notes():
    """INTEGRATION.md's table names, for every function include/bepuhip.h declares, the reference interface it replaces (or says there is none)."""
    import re
    header = open(os.path.join(REPO, "include", "bepuhip.h")).read()
    notes = open(os.path.join(REPO, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(bepuhip_[a-z_0-9]+)\s*\(", header)))
    assert len(names) > 40
    assert [n for n in names if n not in notes] == []


def test_csharp_binding_is_one_text_and_covers_every_entry_point_of_the_header():
    """INTEGRATION.md's C# block and integration/csharp/HipTimestepper.cs used to drift apart. The block IS the file, and the file's DllImport table
    is the one tools/gen_csharp_imports.py generates from include/bepuhip.h (every entry point, parameter for parameter)."""
    import importlib.util
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    source = open(os.path.join(repo, "integration", "csharp", "HipTimestepper.cs")).read()
    doc = open(os.path.join(repo, "INTEGRATION.md")).read()
    blocks = re.findall(r"```csharp\n(.*?)```", doc, flags=re.S)
    assert blocks and blocks[0] == source, "INTEGRATION.md's first csharp block must be integration/csharp/HipTimestepper.cs verbatim"
    spec = importlib.util.spec_from_file_location("gen_csharp_imports", os.path.join(repo, "tools", "gen_csharp_imports.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    generated = gen.block()
    assert generated in source, "the DllImport block of HipTimestepper.cs is stale: regenerate it with tools/gen_csharp_imports.py"
    from bepuphysics2_amd import native
    for name in native.EXPORTED_SYMBOLS:
        assert re.search(r"extern \w+ " + name + r"\(", source), name
    # the calls the timestepper makes exist with the arity it uses them with
    for call, args in re.findall(r"BepuHip\.(bepuhip_\w+)\((.*?)\)\)?;", source):
        declared = re.search(r"extern \w+ " + call + r"\((.*?)\);", source).group(1)

        def arity(text):  # commas at depth 0 (a function-pointer parameter carries commas of its own)
            depth, n = 0, 1 if text.strip() else 0
            for ch in text:
                depth += ch in "(<["
                depth -= ch in ")>]"
                n += ch == "," and depth == 0
            return n

        assert arity(args) == arity(declared), (call, args, declared)
