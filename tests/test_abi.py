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


def test_asm_stores_carry_their_own_wait_states():
    """A VMEM store of more than 64 bits reads its data registers after it has issued; gfx940+ wants two wait states before a VALU instruction overwrites them, and the
    compiler's hazard recognizer does not look inside asm statements (paired records turned into garbage until the stores got an `s_nop 1`). Every asm
    statement of the kernels that stores 96 or 128 bits must therefore end in `s_nop 1` (or more) behind its last store."""
    import re
    csrc = os.path.join(REPO, "bepuphysics2_amd", "csrc")
    wide_store = re.compile(r"(global|buffer|flat|scratch)_store_dwordx[34]")
    found = 0
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".h", ".hip", ".inc")):
            continue
        text = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r"asm\s+volatile\s*\(", text):
            depth, end = 0, m.end() - 1
            for end in range(m.end() - 1, len(text)):  # the statement's closing parenthesis
                depth += text[end] == "("
                depth -= text[end] == ")"
                if depth == 0:
                    break
            statement = text[m.start():end]
            stores = list(wide_store.finditer(statement))
            if not stores:
                continue
            found += 1
            tail = statement[stores[-1].end():]
            nop = re.search(r"s_nop\s+(\d+)", tail)
            assert nop and int(nop.group(1)) >= 1, f"{name}: an asm store of more than 64 bits without two wait states behind it: {statement[:160]!r}"
    assert found >= 2  # store_agent_f4, store_agent_pair
