"""Binary exchange format between this repository and the C# ReferenceDumper (oracle/pin/ReferenceDumper/Program.cs).

Test infrastructure (part of the oracle's pinning kit): lets anybody with a .NET SDK and a bepuphysics2 checkout run OUR seeded scenes through the
REAL `Simulation.Solve` and diff the result against the oracle — the step this container cannot perform (no dotnet), which is why the oracle's header
says "parity unpinned". Little-endian throughout.

scene file:  b"BEPUPIN1"
             i32 body_count, then body_count x 32 f32        (BodyDynamics: orientation xyzw, position xyz_, linear xyz_, angular xyz_, local inverse inertia
                                                               xx yx yy zx zy zz, inverse mass, _ ... ; BodyProperties.cs:318-338)
             f32 gravity[3], linear_damping, angular_damping; i32 angular_integration_mode, allow_substeps_for_unconstrained, integrate_velocity_for_kinematics
             i32 velocity_iterations, substeps, scheduled_count (0 = no VelocityIterationScheduler), scheduled_count x i32
             f32 dt; i32 frames
             i32 batch_count; per batch: i32 type_batch_count; per type batch: i32 type_id, count, bodies, prestep_floats, impulse_floats,
                 then per constraint (in index order): bodies x i32 encoded body references, prestep_floats x f32, impulse_floats x f32
result file: b"BEPUOUT1", i32 body_count, body_count x 32 f32, then per type batch in the scene file's order, per constraint in its order:
             impulse_floats x f32 accumulated impulses, prestep_floats x f32 prestep data (contact depths change during a solve)."""
from __future__ import annotations

import struct

import numpy as np

from bepuphysics2_amd.scene import Scene


def write_scene(path: str, scene: Scene, dt: float, sd, cb, frames: int):
    w = scene.bundle_width
    with open(path, "wb") as f:
        f.write(b"BEPUPIN1")
        f.write(struct.pack("<i", scene.body_count))
        f.write(np.ascontiguousarray(scene.bodies, dtype="<f4").tobytes())
        g = list(cb.gravity)
        f.write(struct.pack("<5f3i", g[0], g[1], g[2], cb.linear_damping, cb.angular_damping, int(cb.angular_integration_mode),
                            int(bool(cb.allow_substeps_for_unconstrained_bodies)), int(bool(cb.integrate_velocity_for_kinematics))))
        its = [int(x) for x in sd.iterations()]
        scheduled = its if sd.velocity_iteration_scheduler is not None else []
        f.write(struct.pack("<3i", int(sd.velocity_iteration_count), int(sd.substep_count), len(scheduled)))
        f.write(struct.pack(f"<{len(scheduled)}i", *scheduled))
        f.write(struct.pack("<fi", dt, frames))
        f.write(struct.pack("<i", len(scene.batches)))
        for batch in scene.batches:
            f.write(struct.pack("<i", len(batch)))
            for tb in batch:
                f.write(struct.pack("<5i", tb.type_id, tb.count, tb.bodies, tb.prestep_floats, tb.impulse_floats))
                refs, pre, acc = tb.refs_lanes(w), tb.prestep_lanes(w), tb.accumulated_lanes(w)
                for i in range(tb.count):
                    f.write(np.asarray(refs[i], "<i4").tobytes())
                    f.write(np.asarray(pre[i], "<f4").tobytes())
                    f.write(np.asarray(acc[i], "<f4").tobytes())


def write_result(path: str, scene: Scene):
    """The result file for a scene already advanced (used with the oracle as a stand-in to test the format end to end)."""
    w = scene.bundle_width
    with open(path, "wb") as f:
        f.write(b"BEPUOUT1")
        f.write(struct.pack("<i", scene.body_count))
        f.write(np.ascontiguousarray(scene.bodies, dtype="<f4").tobytes())
        for batch in scene.batches:
            for tb in batch:
                pre, acc = tb.prestep_lanes(w), tb.accumulated_lanes(w)
                for i in range(tb.count):
                    f.write(np.asarray(acc[i], "<f4").tobytes())
                    f.write(np.asarray(pre[i], "<f4").tobytes())


def read_result(path: str, like: Scene):
    """Returns (bodies [n, 32], [(impulses [count, imf], prestep [count, pf]) per type batch in scene order])."""
    data = open(path, "rb").read()
    if data[:8] != b"BEPUOUT1":
        raise ValueError("not a ReferenceDumper result file")
    (n,) = struct.unpack_from("<i", data, 8)
    if n != like.body_count:
        raise ValueError(f"body count {n} != scene's {like.body_count}")
    off = 12
    bodies = np.frombuffer(data, "<f4", n * 32, off).reshape(n, 32).copy()
    off += n * 128
    per_tb = []
    for batch in like.batches:
        for tb in batch:
            per = tb.impulse_floats + tb.prestep_floats
            block = np.frombuffer(data, "<f4", tb.count * per, off).reshape(tb.count, per).copy()
            off += tb.count * per * 4
            per_tb.append((block[:, :tb.impulse_floats], block[:, tb.impulse_floats:]))
    if off != len(data):
        raise ValueError("trailing bytes in result file")
    return bodies, per_tb
