"""In-tree builds: hipcc for the gfx950 library, g++ for the C++ host mirror. Built artefacts stay next to their sources
(git-ignored, but shipped to the GPU box by gpurun)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Xarch_device", "-fno-slp-vectorize",
             "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value", "-Wno-array-bounds"]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


HIP_COMPILE_FLAGS = [f for f in HIP_FLAGS if f != "-shared"]


def _hip_units(src_dir: str):
    return sorted(f for f in os.listdir(src_dir) if f.endswith(".hip"))


CODE_PAD_BYTES = 4 * 8192  # kCodeTouchMaxSpans x 8 KB (bepu_kernels_common.h)


def check_code_pad(obj: str) -> None:
    """cluster_kernel's code touch reads up to CODE_PAD_BYTES ahead of a wave's PC as data (bepu_cluster_kernel.h, touch_code_span). Every cluster unit therefore ends
    in that much never-executed padding, code_pad_kernel; the compiler lays functions out in instantiation order, which this checks in the built gfx950 code object:
    the pad is the LAST function of .text, long enough, and every cluster_kernel lies before it."""
    llvm = "/opt/rocm/lib/llvm/bin"
    fat, co = obj + ".fat", obj + ".co"
    try:
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                              stderr=subprocess.DEVNULL)
        table = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "-sW", co], text=True)
    finally:
        for f in (fat, co):
            if os.path.exists(f):
                os.remove(f)
    functions = {}
    for line in table.splitlines():
        cols = line.split()
        if len(cols) >= 8 and cols[3] == "FUNC":
            functions[cols[7]] = (int(cols[1], 16), int(cols[2]))
    pads = [(a, n) for name, (a, n) in functions.items() if "code_pad_kernel" in name]
    kernels = [(a, n) for name, (a, n) in functions.items() if "cluster_kernel" in name]
    if len(pads) != 1 or not kernels:
        raise RuntimeError(f"{obj}: expected one code_pad_kernel and at least one cluster_kernel in the gfx950 code object")
    pad_at, pad_bytes = pads[0]
    if pad_bytes < CODE_PAD_BYTES or any(a + n > pad_at for a, n in functions.values() if (a, n) != pads[0]):
        raise RuntimeError(f"{obj}: code_pad_kernel ({pad_bytes} bytes at {pad_at:#x}) is not the last function of the unit or is shorter than {CODE_PAD_BYTES} bytes: "
                           "cluster_kernel's code touch could read past the code object")


def build_hip(force: bool = False, jobs: int = 0) -> str:
    """libbepuhip.so = bepuhip.hip (C ABI, launch-per-batch / stream / per-body kernels) + one translation unit per cluster_kernel register budget
    (bepu_cluster_{hot,wide}_{1024,512}[s|n|c|p].hip). Units are compiled to objects in parallel and linked; an object is rebuilt when any source is newer."""
    from concurrent.futures import ThreadPoolExecutor
    src_dir = os.path.join(_HERE, "csrc")
    obj_dir = os.path.join(src_dir, "build")
    out = os.path.join(src_dir, "libbepuhip.so")
    sources = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h", ".inc"))] + [os.path.join(REPO, "include", "bepuhip.h")]
    units = _hip_units(src_dir)
    objects = [os.path.join(obj_dir, u[:-4] + ".o") for u in units]
    if not force and _newer(out, sources):
        return out
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()

    # read by bepuhip.hip alone: the cluster units do not depend on them (nor on the public header: they see none of its types)
    host_only = {"bepuhip.hip", "bepu_host_state.h", "bepu_cluster_plan.h", "bepu_soft_updates.h", "bepu_transfer_kernels.h", "bepu_colour_kernels.h", "bepuhip.h"}
    cluster_only = {"bepu_cluster_kernel.h", "bepu_cluster_variant.inc"}

    def unit_sources(unit):
        skip = cluster_only if unit == "bepuhip.hip" else host_only
        own = os.path.join(src_dir, unit)
        return [s for s in sources if s == own or (os.path.basename(s) not in skip and not s.endswith(".hip"))]

    def compile_unit(unit_object):
        unit, obj = unit_object
        if not force and _newer(obj, unit_sources(unit)):
            return
        tmp = obj + f".tmp{os.getpid()}"
        subprocess.check_call([hipcc] + HIP_COMPILE_FLAGS + ["-c", "-o", tmp, os.path.join(src_dir, unit)], cwd=src_dir)
        if unit.startswith("bepu_cluster_"):
            check_code_pad(tmp)
        os.replace(tmp, obj)

    with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_unit, zip(units, objects)))
    tmp = out + f".tmp{os.getpid()}"  # link aside, then rename: a concurrent reader never sees a half-written library
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objects, cwd=src_dir)
    os.replace(tmp, out)
    return out


def build_host(force: bool = False) -> str:
    src_dir = os.path.join(_HERE, "host")
    out = os.path.join(src_dir, "libbepuhost.so")
    sources = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".cpp", ".h"))] + [os.path.join(REPO, "include", "bepuhip.h")]
    cpps = [s for s in sources if s.endswith(".cpp")]
    if not cpps:
        return ""
    if not force and _newer(out, sources):
        return out
    tmp = out + f".tmp{os.getpid()}"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-I", os.path.join(REPO, "include"), "-o", tmp] + cpps + ["-ldl"]
    subprocess.check_call(cmd, cwd=src_dir)
    os.replace(tmp, out)
    return out


# Units of the island kernel for the exact type sets of the BASELINE.json scenes (bepu_unit_cache.h): compiled by the library's own bepuhip_prebuild_unit into
# csrc/units/, where a context that asks for its unit (bepuhip_specialise_units) finds it without a compiler run. (type mask, threads budget, split plan, what it is)
BASELINE_UNITS = [
    (0xC0004E4000F8, 1024, 0, "configs[2] / [3]: the ragdoll tube (12 of the 16 hot types), whole-island plans"),
    (0xC0004E4000F8, 768, 1, "bench.py's connected ragdoll crowd: the same types on a split plan at twelve waves (no gain measured: prebuilt so that the bench's setup never waits for a compiler)"),
    (0xF8, 512, 1, "configs[1]: the 100k-box pile (Contact1-4 two-body + Contact4 one-body), split plan at eight waves"),
    (0x20C204FFC000FF, 1024, 0, "bench.py's widened_types leg: the sixteen hot types + seven widened joint types, whole-island plans"),
]


def build_units(jobs: int = 0) -> list:
    """The BASELINE_UNITS, in parallel (a unit is a 40-60 s hipcc run; found in the cache when the sources have not changed). Objects of other source states are removed."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    lib = ctypes.CDLL(os.path.join(_HERE, "csrc", "libbepuhip.so"))
    lib.bepuhip_prebuild_unit.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]
    lib.bepuhip_last_error.restype = ctypes.c_char_p

    def one(unit):
        mask, budget, split, what = unit
        path = ctypes.create_string_buffer(1024)
        status = lib.bepuhip_prebuild_unit(mask, budget, split, path, 1024)
        if status != 0:
            raise RuntimeError(f"unit for {what}: {lib.bepuhip_last_error().decode()}")
        return path.value.decode()

    with ThreadPoolExecutor(max_workers=jobs or len(BASELINE_UNITS)) as pool:
        paths = list(pool.map(one, BASELINE_UNITS))
    units_dir = os.path.join(_HERE, "csrc", "units")
    if os.path.isdir(units_dir) and paths:
        current = os.path.basename(paths[0]).split("_")[1]  # unit_<sources hash>_m<mask>_t<budget>[s].so
        for name in os.listdir(units_dir):
            if name.startswith("unit_") and name.split("_")[1] != current:
                os.remove(os.path.join(units_dir, name))
    return paths


def build_oracle() -> None:
    """Compile oracle/'s C++ restatement (test infrastructure; building the checker is not using it)."""
    subprocess.check_call(["make", "-s"], cwd=os.path.join(REPO, "oracle"))


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_units()
    build_host(force)
    build_oracle()
