"""ContactFused (csrc/bepu_device_constraints.h) — the manifold function with a per-lane contact count that the merged manifold work items of the split plans run —
against Contact<N, TwoBody> for N = 1..4, one and two bodies: the same bits out of warm start, solve and the incremental depth update, on the host
(the header compiled as plain C++ with -ffp-contract=off; the arithmetic is IEEE fp32 on either side). The reference generates the four types from one
template (BepuPhysics/Constraints/Contact/ContactConvexTypes.cs:901-1514); what differs per N is restated per lane here, and this test pins it."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "contact_fused_host")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_contact_fused_equals_typed_manifolds_bit_for_bit(tmp_path):
    exe = str(tmp_path / "fused_host")
    subprocess.check_call([CLANG, "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(SRC, "shim"), "-I", os.path.join(REPO, "bepuphysics2_amd", "csrc"),
                           "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-array-bounds", "-o", exe, os.path.join(SRC, "fused_host.cpp")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe, "4000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout
