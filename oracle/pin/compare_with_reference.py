"""Diffs ReferenceDumper results against the CPU oracle. Usage: python oracle/pin/compare_with_reference.py <dir with *.scene.bin and *.result.bin>

For every case: the oracle advances the same scene the same number of frames; poses, velocities, accumulated impulses and prestep data are compared
bit for bit, and the maximum ULP distance / relative velocity error is printed where they differ. All cases bit-exact => the oracle is pinned
(remove "parity unpinned" from oracle/bepu_oracle.cpp's header, DESIGN.md and this directory's README)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle_ffi  # noqa: E402
import parity_util as pu  # noqa: E402
import pin_format  # noqa: E402
from export_pin_scenes import pin_cases  # noqa: E402

COLS = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]  # orientation, position, linear, angular (the padding lanes are not defined by the reference)


def compare_case(scene, dt, sd, cb, frames, result_path):
    ref_bodies, ref_tbs = pin_format.read_result(result_path, scene)
    s = scene.copy()
    for _ in range(frames):
        oracle_ffi.solve(s, dt, sd, cb)
    out = {"bodies_bit_exact": bool(np.array_equal(ref_bodies[:, COLS].view(np.int32), s.bodies[:, COLS].view(np.int32))),
           "bodies_max_ulp": pu.max_ulp_diff(ref_bodies[:, COLS], s.bodies[:, COLS])}
    vel = [8, 9, 10, 12, 13, 14]
    out["velocity_rel_err"] = float(np.abs(ref_bodies[:, vel] - s.bodies[:, vel]).max() / max(float(np.abs(ref_bodies[:, vel]).max()), 1e-6))
    imp, pre, ulp = True, True, 0
    tbs = [tb for batch in s.batches for tb in batch]
    for tb, (ri, rp) in zip(tbs, ref_tbs):
        oi, op = tb.accumulated_lanes(s.bundle_width), tb.prestep_lanes(s.bundle_width)
        imp &= bool(np.array_equal(ri.view(np.int32), oi.view(np.int32)))
        pre &= bool(np.array_equal(rp.view(np.int32), op.view(np.int32)))
        ulp = max(ulp, pu.max_ulp_diff(ri, oi))
    out.update(impulses_bit_exact=imp, prestep_bit_exact=pre, impulses_max_ulp=ulp)
    if not (out["bodies_bit_exact"] and imp and pre):
        # Second opinions for a case that differs: oracle/wide (the independent AOSOA transcription), and — the one difference that is EXPECTED on an x86 host, for the four
        # types built on MathHelper.FastReciprocal[SquareRoot] — oracle/wide with the helpers as vrcpps / vrsqrtps (bit-exact only on the CPU vendor the dumper ran on).
        try:
            import wide_ffi
            for variant in ("", "rcpx86", "zerominus"):
                w = scene.copy()
                for _ in range(frames):
                    wide_ffi.solve(w, dt, sd, cb, variant=variant)
                out[f"wide{'_' + variant if variant else ''}_bodies_bit_exact"] = bool(np.array_equal(ref_bodies[:, COLS].view(np.int32), w.bodies[:, COLS].view(np.int32)))
        except Exception as e:  # noqa: BLE001 — the second opinion is optional (no AVX2 host)
            out["wide"] = f"unavailable: {e}"
    return out


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "pin_scenes"
    worst = differing = total = 0
    for name, scene, dt, sd, cb, frames in pin_cases():
        result = os.path.join(d, name + ".result.bin")
        if not os.path.exists(result):
            print(f"{name}: no result file (run ReferenceDumper on {name}.scene.bin)")
            worst = max(worst, 1)
            continue
        m = compare_case(scene, dt, sd, cb, frames, result)
        ok = m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"]
        print(f"{name}: {'BIT-EXACT' if ok else 'DIFFERS'} {m}")
        worst = max(worst, 0 if ok else 2)
        differing += 0 if ok else 1
        total += 1
    print(f"VERDICT: {total - differing} of {total} cases bit-exact against the reference" + (" - the oracle is PINNED (remove 'parity unpinned' from oracle/bepu_oracle.cpp, "
          "DESIGN.md and this directory's README)" if worst == 0 else "; cases named *CenterDistance*, *AreaConstraint*, *VolumeConstraint* are expected to differ on x86 hosts "
          "(FastReciprocal: see wide_rcpx86_bodies_bit_exact above) - everything else that differs is a finding"))
    return worst


if __name__ == "__main__":
    sys.exit(main())
