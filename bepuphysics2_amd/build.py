"""In-tree builds: hipcc for the gfx950 library, g++ for the C++ host mirror. Built artefacts stay next to their sources
(git-ignored, but shipped to the GPU box by gpurun)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
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


def build_hip(force: bool = False) -> str:
    src_dir = os.path.join(_HERE, "csrc")
    out = os.path.join(src_dir, "libbepuhip.so")
    sources = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h"))] + [os.path.join(REPO, "include", "bepuhip.h")]
    if not force and _newer(out, sources):
        return out
    tmp = out + f".tmp{os.getpid()}"  # build aside, then rename: a concurrent reader never sees a half-written library
    cmd = [_hipcc()] + HIP_FLAGS + ["-o", tmp, os.path.join(src_dir, "bepuhip.hip")]
    subprocess.check_call(cmd, cwd=src_dir)
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


def build_oracle() -> None:
    """Compile oracle/'s C++ restatement (test infrastructure; building the checker is not using it)."""
    subprocess.check_call(["make", "-s"], cwd=os.path.join(REPO, "oracle"))


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_host(force)
    build_oracle()
