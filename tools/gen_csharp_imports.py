"""Generates the DllImport block of integration/csharp/HipTimestepper.cs from include/bepuhip.h, so that the C# binding cannot drift from the header:
    python tools/gen_csharp_imports.py            prints the block
tests/test_abi.py compares it with what the .cs file holds."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = [
    (r"^bepuhip_ctx\*\*$", "IntPtr*"), (r"^bepuhip_ctx\*$", "IntPtr"), (r"^const bepuhip_config\*$", "BepuHipConfig*"), (r"^const bepuhip_integrator\*$", "BepuHipIntegrator*"),
    (r"^const bepuhip_collidable\*$", "BepuHipCollidable*"), (r"^bepuhip_predicted_bounds\*$", "BepuHipPredictedBounds*"), (r"^const bepuhip_compound_child\*$", "BepuHipCompoundChild*"),
    (r"^const bepuhip_structural_op\*$", "BepuHipStructuralOp*"), (r"^const bepuhip_velocity_model\*$", "BepuHipVelocityModel*"), (r"^const bepuhip_row_transfer\*$", "BepuHipRowTransfer*"),
    (r"^bepuhip_exchange_fn$", "delegate* unmanaged[Cdecl]<void*, int, int, int>"), (r"^bepuhip_substep_fn$", "delegate* unmanaged[Cdecl]<void*, int, void>"),
    (r"^(const )?int32_t\*$", "int*"), (r"^(const )?float\*$", "float*"), (r"^(const )?void\*$", "void*"), (r"^void\*\*$", "void**"), (r"^uint8_t\*$", "byte*"),
    (r"^(const )?uint32_t\*$", "uint*"), (r"^uint64_t\*$", "ulong*"), (r"^int64_t\*$", "long*"), (r"^int32_t$", "int"), (r"^int64_t$", "long"), (r"^uint64_t$", "ulong"), (r"^char\*$", "byte*"), (r"^float$", "float"),
]


def camel(name: str) -> str:
    parts = name.split("_")
    return parts[0] + "".join(p.capitalize() for p in parts[1:])


def declarations():
    text = open(os.path.join(REPO, "include", "bepuhip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    for m in re.finditer(r"(const char\*|int32_t)\s+(bepuhip_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        out = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                ctype, pname = p.rsplit(" ", 1) if not p.endswith("*") else (p, "arg")
                while pname.startswith("*"):
                    ctype, pname = ctype + "*", pname[1:]
                ctype = ctype.replace(" *", "*").strip()
                for pattern, cs in TYPES:
                    if re.match(pattern, ctype):
                        out.append(f"{cs} {camel(pname)}")
                        break
                else:
                    raise SystemExit(f"no C# type for '{ctype}' in {name}")
        yield ("IntPtr" if ret.startswith("const char") else "int"), name, out


def block() -> str:
    lines = []
    for ret, name, params in declarations():
        lines.append(f"    [DllImport(Lib)] public static extern {ret} {name}({', '.join(params)});")
    return "\n".join(lines)


if __name__ == "__main__":
    print(block())
