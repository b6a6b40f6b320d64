"""Scans the gfx950 code objects of the built library for the store-data hazard the compiler cannot see inside asm statements.

A VMEM store of more than 64 bits reads its data registers AFTER it has issued; gfx940+ needs two wait states before a VALU instruction overwrites them. LLVM's hazard
recognizer inserts them behind the stores it emits itself and does not look inside asm statements — round 4's paired records were garbage until the kernels' asm stores
got their own `s_nop 1`. The source-text check that pinned this could not see a NEW asm statement with another mnemonic, nor a store the compiler schedules
differently: this scan works on what actually runs. For every `*_store_dwordx3/x4` (global, flat, buffer, scratch) in the disassembly it walks the following
instructions until two wait states have passed (an instruction is one wait state, `s_nop N` is N + 1) and reports any VALU instruction that writes one of the store's
data VGPRs in between; a branch ends the walk for that store (its targets are not followed: a hazard across a taken branch costs the branch's own wait states).

    python tools/check_store_hazard.py [object files ...]      default: every object of bepuphysics2_amd/csrc/build
Exit status 1 when a hazard is found. `scan_text` is what tests/test_store_hazard.py calls (also on a deliberately broken build)."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
WIDE_STORE = re.compile(r"^(global|flat|buffer|scratch)_store_(dwordx[34]|b96|b128)\b")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def registers(operand: str):
    m = REG.search(operand)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def store_data_registers(mnemonic: str, operands):
    """global/flat/scratch: addr, data, saddr...; buffer: data, vaddr, srsrc, soffset — the data operand is the widest VGPR tuple."""
    best = set()
    for op in operands:
        regs = registers(op)
        if len(regs) >= 3 and len(regs) > len(best):
            best = regs
    return best


def valu_written_registers(mnemonic: str, operands):
    if not mnemonic.startswith("v_") or not operands:
        return set()
    if mnemonic.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
        return set()  # write SGPRs / VCC / EXEC
    written = registers(operands[0]) if operands[0].lstrip().startswith("v") else set()
    if mnemonic.startswith(("v_swap", "v_mad_u64", "v_mad_i64")) and len(operands) > 1:  # two destinations
        written |= registers(operands[1]) if mnemonic.startswith("v_swap") else set()
    return written


def parse(text: str):
    """[(function, [(mnemonic, [operands], raw line)])] from llvm-objdump -d output."""
    functions, current = [], None
    for line in text.splitlines():
        head = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if head:
            current = (head.group(1), [])
            functions.append(current)
            continue
        body = line.split("//")[0].strip()
        if not body or current is None:
            continue
        parts = body.split(None, 1)
        mnemonic = parts[0]
        if not re.match(r"^[a-z_0-9]+$", mnemonic):
            continue
        operands = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        current[1].append((mnemonic, operands, body))
    return functions


def scan_text(text: str):
    """Hazards in a disassembly: [(function, store line, offending line, wait states between them)]; and the number of wide stores looked at."""
    hazards, stores = [], 0
    for name, instructions in parse(text):
        for at, (mnemonic, operands, raw) in enumerate(instructions):
            if not WIDE_STORE.match(mnemonic):
                continue
            data = store_data_registers(mnemonic, operands)
            if not data:
                continue
            stores += 1
            waited = 0
            for later_mnemonic, later_operands, later_raw in instructions[at + 1:]:
                if waited >= 2:
                    break
                if later_mnemonic.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                if valu_written_registers(later_mnemonic, later_operands) & data:
                    hazards.append((name, raw, later_raw, waited))
                    break
                if later_mnemonic == "s_nop":
                    waited += int(later_operands[0], 0) + 1 if later_operands else 1
                else:
                    waited += 1
    return hazards, stores


def disassemble(obj: str) -> str:
    """The gfx950 code object inside a host object / shared library built by hipcc, disassembled."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                              stderr=subprocess.DEVNULL)
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], text=True)


def main(argv):
    objects = argv or sorted(os.path.join(REPO, "bepuphysics2_amd", "csrc", "build", f) for f in os.listdir(os.path.join(REPO, "bepuphysics2_amd", "csrc", "build")) if f.endswith(".o"))
    bad = 0
    for obj in objects:
        hazards, stores = scan_text(disassemble(obj))
        print(f"{os.path.basename(obj)}: {stores} stores of more than 64 bits, {len(hazards)} hazards")
        for name, store, offender, waited in hazards:
            print(f"    {name[:80]}: `{store}` then, {waited} wait state(s) later, `{offender}`")
        bad += len(hazards)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
