"""Algorithmic bytes per constraint per stage (SURVEY.md §8d / BASELINE.md §3) — the numerator of ``roofline.achieved``.

Solve     = P + 2A + R + sum_bodies(fields gathered by the Solve access filter + velocity written)
WarmStart = P +  A + R + sum_bodies(fields gathered by the WarmStart access filter + velocity written)
Incremental (contacts) = 16n + 12 (normal) [+12 OffsetB] read + 4n write + R + 24 per body
Field sizes: position 12, orientation 16, linear 12, angular 12, inverse inertia tensor 24, inverse mass 4.
Cache hits on body data do not reduce the algorithmic byte count."""
from __future__ import annotations

from .scene import TYPE_TABLE, Scene

POS, ORI, LIN, ANG, INERTIA = 1, 2, 4, 8, 16
ALL, NO_POSITION, NO_POSE, ONLY_ANGULAR, ONLY_ANGULAR_NO_POSE, ONLY_LINEAR = 31, 30, 28, 26, 24, 21

# (warm-start A, warm-start B, solve A, solve B) access filters, as in the reference's TypeProcessor declarations.
ACCESS = {
    "Contact": (NO_POSE, NO_POSE, NO_POSE, NO_POSE),
    "BallSocket": (NO_POSITION, NO_POSITION, ALL, ALL),
    "AngularHinge": (ONLY_ANGULAR, ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR, ONLY_ANGULAR),
    "SwingLimit": (ONLY_ANGULAR,) * 4, "TwistServo": (ONLY_ANGULAR,) * 4, "TwistLimit": (ONLY_ANGULAR,) * 4,
    "AngularMotor": (ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR, ONLY_ANGULAR_NO_POSE),
    "SwivelHinge": (NO_POSITION, NO_POSITION, ALL, ALL), "Hinge": (NO_POSITION, NO_POSITION, ALL, ALL),
    "Weld": (NO_POSITION, NO_POSE, ALL, ALL),
    "AngularSwivelHinge": (ONLY_ANGULAR,) * 4, "TwistMotor": (ONLY_ANGULAR,) * 4, "AngularServo": (ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR, ONLY_ANGULAR),
    "DistanceServo": (ALL,) * 4, "DistanceLimit": (ALL,) * 4, "AngularAxisMotor": (ONLY_ANGULAR, ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR, ONLY_ANGULAR),
    "OneBodyAngularServo": (ONLY_ANGULAR, 0, ONLY_ANGULAR, 0), "OneBodyAngularMotor": (ONLY_ANGULAR_NO_POSE, 0, ONLY_ANGULAR, 0),
    "OneBodyLinearServo": (ALL, 0, ALL, 0), "OneBodyLinearMotor": (NO_POSITION, 0, NO_POSITION, 0),
    "BallSocketMotor": (29, ALL, ALL, ALL), "BallSocketServo": (NO_POSITION, NO_POSITION, ALL, ALL),
    "PointOnLineServo": (ALL,) * 4, "LinearAxisServo": (ALL,) * 4, "LinearAxisMotor": (ALL,) * 4, "LinearAxisLimit": (ALL,) * 4,
    "AngularAxisGearMotor": (ONLY_ANGULAR, ONLY_ANGULAR_NO_POSE, ONLY_ANGULAR, ONLY_ANGULAR_NO_POSE),
    "CenterDistanceConstraint": (ONLY_LINEAR,) * 4, "CenterDistanceLimit": (ONLY_LINEAR,) * 4,
    "AreaConstraint": (ONLY_LINEAR,) * 4, "VolumeConstraint": (ONLY_LINEAR,) * 4,  # every body of the three / four uses the same filter
}


def _body_bytes(mask: int) -> int:
    read = (12 if mask & POS else 0) + (16 if mask & ORI else 0) + (12 if mask & LIN else 0) + (12 if mask & ANG else 0)
    if mask & INERTIA:
        if not mask & ANG and not mask & ORI:
            read += 4   # AccessOnlyLinear: the inverse mass alone
        else:
            read += 28 if mask & LIN else 24  # angular-only filters do not need the inverse mass
    write = (12 if mask & LIN else 0) + (12 if mask & ANG else 0)
    return read + write


def stage_bytes(type_id: int):
    """(warm_start_bytes, solve_bytes, incremental_bytes) per constraint."""
    nb, pf, imf, name = TYPE_TABLE[type_id]
    acc = ACCESS["Contact"] if name.startswith("Contact") else ACCESS[name]
    p, a, r = pf * 4, imf * 4, nb * 4
    ws = p + a + r + _body_bytes(acc[0]) + (_body_bytes(acc[1]) * (nb - 1) if nb >= 2 else 0)
    sv = p + 2 * a + r + _body_bytes(acc[2]) + (_body_bytes(acc[3]) * (nb - 1) if nb >= 2 else 0)
    inc = 0
    if name.startswith("Contact"):
        n = int(name[7])
        if "Nonconvex" in name:  # every contact carries its own normal: 28 B read + 4 B depth written per contact
            inc = 28 * n + (12 if nb == 2 else 0) + 4 * n + r + 24 * nb
        else:
            inc = 16 * n + 12 + (12 if nb == 2 else 0) + 4 * n + r + 24 * nb
    return ws, sv, inc


INTEGRATE_BYTES_PER_BODY = 96 + 28 + 28 + 24  # per integrating body per substep: 96 read, pose + world inertia + velocity written
FINAL_BYTES_PER_BODY = 160                   # 128 read + 32 pose write


def scene_stage_bytes(scene: Scene):
    """Total algorithmic bytes of ONE warm-start pass, ONE solve pass and ONE incremental pass over the whole scene."""
    ws = sv = inc = 0
    for b in scene.batches:
        for tb in b:
            w, s, i = stage_bytes(tb.type_id)
            ws += w * tb.count
            sv += s * tb.count
            inc += i * tb.count
    return ws, sv, inc
