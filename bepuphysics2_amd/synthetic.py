"""Synthetic constraint descriptions for scenes with the widened constraint types: seeded prestep lanes per joint type (used by the parity tests' scene generators,
tests/small_scenes.py) and the "rigs" scene of bench.py's widened_types leg."""
from __future__ import annotations

import math

import numpy as np

from .scene import TYPE_TABLE

TWO_PI = 6.283185307179586
FLOAT_MAX = float(np.finfo(np.float32).max)


def unit(rng, n=3):
    v = rng.normal(size=n)
    return (v / np.linalg.norm(v)).astype(np.float32)


def rand_quat(rng, spread=1.0):
    q = rng.normal(size=4) * np.array([spread, spread, spread, 1.0])
    return (q / np.linalg.norm(q)).astype(np.float32)


def spring(frequency, damping_ratio):
    # SpringSettings(frequency, dampingRatio): AngularFrequency = f * TwoPi, TwiceDampingRatio = 2 * zeta (SpringSettings.cs:73-78)
    return [np.float32(frequency) * np.float32(TWO_PI), np.float32(damping_ratio) * np.float32(2)]


def joint_prestep(rng, type_id):
    name = TYPE_TABLE[type_id][3]
    sp = spring(15.0, 1.0)
    if name == "BallSocket":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + sp
    if name == "AngularHinge":
        return list(unit(rng)) + list(unit(rng)) + sp
    if name == "SwingLimit":
        return list(unit(rng)) + list(unit(rng)) + [math.cos(rng.uniform(0.2, 2.5))] + sp
    if name == "TwistServo":
        servo = [FLOAT_MAX, 0.0, FLOAT_MAX] if rng.random() < 0.5 else [rng.uniform(1, 5), rng.uniform(0, 0.5), rng.uniform(10, 1000)]
        return list(rand_quat(rng)) + list(rand_quat(rng)) + [rng.uniform(-0.5, 0.5)] + sp + servo
    if name == "TwistLimit":
        a = rng.uniform(0.1, 1.5)
        return list(rand_quat(rng)) + list(rand_quat(rng)) + [-a, a] + sp
    if name == "AngularMotor":
        settings = [FLOAT_MAX, 1.0 / 0.01] if rng.random() < 0.5 else [rng.uniform(1, 100), rng.uniform(1, 200)]
        return list(rng.uniform(-0.2, 0.2, 3)) + settings
    def servo():  # ServoSettings{MaximumSpeed, BaseSpeed, MaximumForce}: unlimited half of the time, as the ragdoll's twist servos are
        return [FLOAT_MAX, 0.0, FLOAT_MAX] if rng.random() < 0.5 else [rng.uniform(1, 5), rng.uniform(0, 0.5), rng.uniform(10, 1000)]

    def motor():  # MotorSettings{MaximumForce, Damping}
        return [FLOAT_MAX, 1.0 / 0.01] if rng.random() < 0.5 else [rng.uniform(1, 100), rng.uniform(1, 200)]

    if name == "AngularSwivelHinge":
        return list(unit(rng)) + list(unit(rng)) + sp
    if name == "TwistMotor":
        return list(unit(rng)) + list(unit(rng)) + [rng.uniform(-1, 1)] + motor()
    if name == "AngularServo":
        return list(rand_quat(rng)) + sp + servo()
    if name == "DistanceServo":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + [rng.uniform(0.5, 3.0)] + servo() + sp
    if name == "DistanceLimit":
        lo = rng.uniform(0.2, 2.0)
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + [lo, lo + rng.uniform(0.1, 2.0)] + sp
    if name == "AngularAxisMotor":
        return list(unit(rng)) + [rng.uniform(-1, 1)] + motor()
    if name == "OneBodyAngularServo":
        return list(rand_quat(rng)) + sp + servo()
    if name == "OneBodyAngularMotor":
        return list(rng.uniform(-1, 1, 3)) + motor()
    if name == "OneBodyLinearServo":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-3, 3, 3)) + sp + servo()
    if name == "OneBodyLinearMotor":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-1, 1, 3)) + motor()
    if name == "BallSocketMotor":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-1, 1, 3)) + motor()
    if name == "BallSocketServo":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + sp + servo()
    if name == "PointOnLineServo":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + list(unit(rng)) + servo() + sp
    if name == "LinearAxisServo":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + list(unit(rng)) + [rng.uniform(-1, 1)] + servo() + sp
    if name == "LinearAxisMotor":
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + list(unit(rng)) + [rng.uniform(-1, 1)] + motor()
    if name == "LinearAxisLimit":
        lo = rng.uniform(-2.0, 1.0)
        return list(rng.uniform(-0.4, 0.4, 3)) + list(rng.uniform(-0.4, 0.4, 3)) + list(unit(rng)) + [lo, lo + rng.uniform(0.1, 2.0)] + sp
    if name == "AngularAxisGearMotor":
        return list(unit(rng)) + [rng.uniform(0.25, 3.0)] + motor()
    if name == "CenterDistanceConstraint":
        return [rng.uniform(0.5, 3.0)] + sp
    if name == "CenterDistanceLimit":
        lo = rng.uniform(0.2, 2.0)
        return [lo, lo + rng.uniform(0.1, 2.0)] + sp
    if name == "AreaConstraint":
        return [rng.uniform(0.5, 6.0)] + sp       # TargetScaledArea = 2 x area
    if name == "VolumeConstraint":
        return [rng.uniform(-6.0, 6.0)] + sp      # TargetScaledVolume = 6 x signed volume
    if name == "Weld":
        return list(rng.uniform(-0.5, 0.5, 3)) + list(rand_quat(rng)) + sp
    if name in ("SwivelHinge", "Hinge"):
        return list(rng.uniform(-0.4, 0.4, 3)) + list(unit(rng)) + list(rng.uniform(-0.4, 0.4, 3)) + list(unit(rng)) + sp
    raise KeyError(name)



# The bench scene's joint types other than BallSocket -> widened types with the same body count (AngularSwivelHinge, DistanceLimit, AngularServo, TwistMotor,
# AngularAxisMotor, Weld, BallSocketServo): same bodies, same constraint graph, same batches, type batches as long as the headline's.
RIG_REMAP = {23: 24, 25: 34, 26: 29, 27: 28, 30: 41, 47: 31, 46: 53}


def rig_scene(ragdolls: int, seed: int = 3):
    from .hostlib import HostSimulation
    from .scene import TypeBatchData, to_aosoa
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 0, 5)
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    rng = np.random.default_rng(seed)
    w = scene.bundle_width
    for batch in scene.batches:
        for i, tb in enumerate(batch):
            if tb.type_id not in RIG_REMAP:
                continue
            t = RIG_REMAP[tb.type_id]
            lanes = np.asarray([joint_prestep(rng, t) for _ in range(tb.count)], dtype=np.float32)
            batch[i] = TypeBatchData(t, tb.count, tb.body_refs, to_aosoa(lanes, w), np.zeros(((tb.count + w - 1) // w) * w * TYPE_TABLE[t][2], dtype=np.float32))
    return scene, sd
