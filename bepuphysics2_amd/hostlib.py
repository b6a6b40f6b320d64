"""ctypes binding of the C++ host mirror (host/libbepuhost.so): Bodies / Solver.Add / batch colouring / scene recipes /
HipTimestepper. The buffers it exposes are in the reference's own layouts (AoS BodyDynamics, AOSOA type batches, W=8)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

from .scene import Scene, SolveDescription, TypeBatchData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libbepuhost.so")
_lib: Optional[C.CDLL] = None


class TypeBatchView(C.Structure):
    _fields_ = [("type_id", C.c_int32), ("count", C.c_int32), ("bodies", C.c_int32), ("prestep_floats", C.c_int32), ("impulse_floats", C.c_int32),
                ("bundle_count", C.c_int32), ("refs", C.POINTER(C.c_int32)), ("prestep", C.POINTER(C.c_float)), ("accumulated", C.POINTER(C.c_float))]


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: run __graft_entry__.build() first")
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    lib.bepuhost_last_error.restype = C.c_char_p
    lib.bepuhost_simulation_create.restype = vp
    lib.bepuhost_simulation_create.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int, C.c_int]
    lib.bepuhost_scene_create.restype = vp
    lib.bepuhost_scene_create.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint32]
    lib.bepuhost_simulation_destroy.argtypes = [vp]
    lib.bepuhost_add_body.argtypes = [vp, vp, vp, vp]
    lib.bepuhost_add_body.restype = i32
    lib.bepuhost_add_constraint.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    lib.bepuhost_add_constraint.restype = i32
    lib.bepuhost_validate.argtypes = [vp]
    for name in ("body_count", "handle_capacity", "constraint_count", "batch_count", "kinematic_count"):
        getattr(lib, "bepuhost_" + name).argtypes = [vp]
        getattr(lib, "bepuhost_" + name).restype = i32
    lib.bepuhost_type_batch_count.argtypes = [vp, C.c_int]
    lib.bepuhost_type_batch_count.restype = i32
    lib.bepuhost_bodies_ptr.argtypes = [vp]
    lib.bepuhost_bodies_ptr.restype = C.POINTER(C.c_float)
    for name in ("index_to_handle", "handle_to_index", "kinematic_handles"):
        getattr(lib, "bepuhost_" + name).argtypes = [vp]
        getattr(lib, "bepuhost_" + name).restype = C.POINTER(C.c_int32)
    lib.bepuhost_type_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(TypeBatchView)]
    lib.bepuhost_solve_description.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.bepuhost_prepare_flags.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, vp]
    lib.bepuhost_prepare_flags.restype = i32
    lib.bepuhost_attach_hip_timestepper.argtypes = [vp, C.c_char_p, C.c_int]
    lib.bepuhost_attach_hip_timestepper.restype = i32
    lib.bepuhost_timestep.argtypes = [vp, C.c_float]
    lib.bepuhost_timestep.restype = i32
    lib.bepuhost_remove_constraint.argtypes = [vp, i32]
    lib.bepuhost_remove_constraint.restype = i32
    lib.bepuhost_type_batch_handles.argtypes = [vp, C.c_int, C.c_int]
    lib.bepuhost_type_batch_handles.restype = C.POINTER(C.c_int32)
    lib.bepuhost_timestepper_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.bepuhost_timestepper_stats.restype = None
    lib.bepuhost_timestepper_mode.argtypes = [vp, C.c_int]
    lib.bepuhost_timestepper_mode.restype = i32
    lib.bepuhost_apply_description.argtypes = [vp, i32, vp]
    lib.bepuhost_apply_description.restype = i32
    lib.bepuhost_set_accumulated_impulses.argtypes = [vp, i32, vp]
    lib.bepuhost_set_accumulated_impulses.restype = i32
    lib.bepuhost_constraint_location.argtypes = [vp, i32, vp]
    lib.bepuhost_constraint_location.restype = i32
    lib.bepuhost_resident_stats.argtypes = [vp, vp]
    lib.bepuhost_resident_stats.restype = i32
    lib.bepuhost_timestepper_replan_interval.argtypes = [vp, C.c_int]
    lib.bepuhost_timestepper_replan_interval.restype = i32
    lib.bepuhost_timestepper_read_back_contact_depths.argtypes = [vp, C.c_int]
    lib.bepuhost_timestepper_read_back_contact_depths.restype = i32
    lib.bepuhost_diff_type_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.POINTER(i32)]
    lib.bepuhost_diff_type_batch.restype = i32
    lib.bepuhost_diff_type_batch_identities.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.POINTER(i32), vp]
    lib.bepuhost_diff_type_batch_identities.restype = i32
    _lib = lib
    return lib


def _err(lib) -> str:
    return lib.bepuhost_last_error().decode("utf-8", "replace")


def _np_from(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)


class HostSimulation:
    """Owns a C++ ``bepu::Simulation`` (Simulation.cs:106-326 mirror)."""

    def __init__(self, handle, lib):
        self.lib, self.h = lib, handle

    @classmethod
    def create(cls, gravity=(0, -10, 0), linear_damping=0.03, angular_damping=0.03, velocity_iterations=1, substeps=1) -> "HostSimulation":
        lib = load_library()
        g = (C.c_float * 3)(*gravity)
        h = lib.bepuhost_simulation_create(g, linear_damping, angular_damping, velocity_iterations, substeps)
        if not h:
            raise ValueError(_err(lib))  # SolveDescription validation (SolveDescription.cs:42-47)
        return cls(h, lib)

    @classmethod
    def scene(cls, name: str, a: int = 0, b: int = 0, c: int = 0, seed: int = 5) -> "HostSimulation":
        lib = load_library()
        h = lib.bepuhost_scene_create(name.encode(), a, b, c, seed)
        if not h:
            raise ValueError(_err(lib))
        return cls(h, lib)

    def close(self):
        if self.h:
            self.lib.bepuhost_simulation_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_body(self, position, orientation, linear, angular, inverse_inertia, inverse_mass) -> int:
        pose = np.asarray(list(position) + list(orientation), dtype=np.float32)
        vel = np.asarray(list(linear) + list(angular), dtype=np.float32)
        inertia = np.asarray(list(inverse_inertia) + [inverse_mass], dtype=np.float32)
        return self.lib.bepuhost_add_body(self.h, pose.ctypes.data, vel.ctypes.data, inertia.ctypes.data)

    def add_constraint(self, type_id: int, body_handles, prestep_lane) -> int:
        hs = np.asarray(body_handles, dtype=np.int32)
        lane = np.asarray(prestep_lane, dtype=np.float32)
        r = self.lib.bepuhost_add_constraint(self.h, type_id, hs.ctypes.data, hs.size, lane.ctypes.data)
        if r < 0:
            raise ValueError(_err(self.lib))
        return r

    def remove_constraint(self, constraint_handle: int):
        """Solver.Remove(handle): swap-with-last inside its type batch (TypeProcessor.cs:634-731)."""
        if self.lib.bepuhost_remove_constraint(self.h, int(constraint_handle)) != 0:
            raise ValueError(_err(self.lib))

    def apply_description(self, constraint_handle: int, prestep_lane):
        """Solver.ApplyDescription(handle, description) (Solver.cs:1162-1185): the description's fields over the constraint's prestep lane, in place."""
        lane = np.ascontiguousarray(prestep_lane, dtype=np.float32)
        if self.lib.bepuhost_apply_description(self.h, int(constraint_handle), lane.ctypes.data) != 0:
            raise ValueError(_err(self.lib))

    def set_accumulated_impulses(self, constraint_handle: int, impulse_lane):
        """The constraint's accumulated impulses written in place (what IslandAwakener.cs:388-400 does when it restores a sleeping island's constraints)."""
        lane = np.ascontiguousarray(impulse_lane, dtype=np.float32)
        if self.lib.bepuhost_set_accumulated_impulses(self.h, int(constraint_handle), lane.ctypes.data) != 0:
            raise ValueError(_err(self.lib))

    def constraint_location(self, constraint_handle: int):
        """(batch index, type id, index in type batch) of a live constraint (Solver.HandleToConstraint), or None."""
        out = (C.c_int32 * 3)()
        if self.lib.bepuhost_constraint_location(self.h, int(constraint_handle), out) != 0:
            return None
        return int(out[0]), int(out[1]), int(out[2])

    def resident_stats(self):
        """(structural operations the attached HipTimestepper's diffs emitted, bundles of joint prestep data / impulses it found changed and sent, the context's schedule)."""
        out = (C.c_int64 * 3)()
        if self.lib.bepuhost_resident_stats(self.h, out) != 0:
            raise RuntimeError(_err(self.lib))
        return int(out[0]), int(out[1]), int(out[2])

    def replan_interval(self, frames: int):
        """Frames a context may spend off its island plan (structural updates the plan could not absorb) before the attached HipTimestepper calls bepuhip_replan (default 30)."""
        if self.lib.bepuhost_timestepper_replan_interval(self.h, int(frames)) != 0:
            raise RuntimeError(_err(self.lib))

    def read_back_contact_depths(self, on: bool = True):
        if self.lib.bepuhost_timestepper_read_back_contact_depths(self.h, int(on)) != 0:
            raise RuntimeError(_err(self.lib))

    def constraint_handles(self, predicate=lambda type_id: True):
        """Handles of the live constraints whose type id satisfies ``predicate`` (TypeBatch.IndexToHandle), in batch / type batch / index order."""
        out = []
        for b in range(self.lib.bepuhost_batch_count(self.h)):
            for t in range(self.lib.bepuhost_type_batch_count(self.h, b)):
                v = TypeBatchView()
                self.lib.bepuhost_type_batch(self.h, b, t, C.byref(v))
                if v.count and predicate(v.type_id):
                    out.extend(_np_from(self.lib.bepuhost_type_batch_handles(self.h, b, t), v.count, np.int32).tolist())
        return out

    def timestepper_stats(self):
        """(full topology uploads, structural-log replays) of the attached HipTimestepper."""
        a, b = C.c_int32(), C.c_int32()
        self.lib.bepuhost_timestepper_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def validate(self):
        if self.lib.bepuhost_validate(self.h) != 0:
            raise AssertionError(_err(self.lib))

    def solve_description(self) -> SolveDescription:
        it, ss = C.c_int32(), C.c_int32()
        self.lib.bepuhost_solve_description(self.h, C.byref(it), C.byref(ss))
        return SolveDescription(it.value, ss.value)

    def export(self) -> Scene:
        """Copy the simulation's buffers (reference layouts) into a Scene."""
        lib, h = self.lib, self.h
        n = lib.bepuhost_body_count(h)
        bodies = _np_from(lib.bepuhost_bodies_ptr(h), n * 32, np.float32).reshape(n, 32)
        i2h = _np_from(lib.bepuhost_index_to_handle(h), n, np.int32)
        h2i = _np_from(lib.bepuhost_handle_to_index(h), lib.bepuhost_handle_capacity(h), np.int32)
        batches: List[List[TypeBatchData]] = []
        for b in range(lib.bepuhost_batch_count(h)):
            tbs = []
            for t in range(lib.bepuhost_type_batch_count(h, b)):
                v = TypeBatchView()
                lib.bepuhost_type_batch(h, b, t, C.byref(v))
                w = 8
                tbs.append(TypeBatchData(v.type_id, v.count, _np_from(v.refs, v.bundle_count * v.bodies * w, np.int32),
                                         _np_from(v.prestep, v.bundle_count * v.prestep_floats * w, np.float32),
                                         _np_from(v.accumulated, v.bundle_count * v.impulse_floats * w, np.float32)))
            batches.append(tbs)
        kin = _np_from(lib.bepuhost_kinematic_handles(h), lib.bepuhost_kinematic_count(h), np.int32)
        return Scene(bodies, i2h, h2i, batches, kin, 8)

    def prepare_flags(self, scene: Scene):
        words = (int(scene.handle_to_index.size) + 63) // 64 or 1
        merged = np.zeros(words, dtype=np.uint64)
        cap = sum(((tb.count + 63) // 64) * tb.bodies for b in scene.batches[1:] for tb in b) + 1
        flags = np.zeros(cap, dtype=np.uint64)
        coarse = np.zeros(max(sum(len(b) for b in scene.batches), 1), dtype=np.uint8)
        if self.lib.bepuhost_prepare_flags(self.h, merged.ctypes.data, words, flags.ctypes.data, cap, coarse.ctypes.data) != 0:
            raise RuntimeError("prepare_flags capacity")
        return merged, flags[:cap - 1], coarse

    def attach_hip_timestepper(self, device: int = 0):
        from .native import LIB_PATH as HIP_LIB
        if self.lib.bepuhost_attach_hip_timestepper(self.h, HIP_LIB.encode(), device) != 0:
            raise RuntimeError(_err(self.lib))

    def timestepper_mode(self, mode: int):
        """0: the attached HipTimestepper replays the solver's structural log; 1: it diffs the type batches against last frame's copy (public API only) and sends the
        frame's changes in one bepuhip_apply_structural_ops call; 2: the resident frame of integration/csharp/HipTimestepper.cs — the diff in two phases, what the host
        rewrote in place (Solver.ApplyDescription, restored impulses, reused handles) found by comparison against a shadow of the device's rows and sent by bundle range,
        everything back asynchronously behind the solve."""
        if self.lib.bepuhost_timestepper_mode(self.h, int(mode)) != 0:
            raise RuntimeError(_err(self.lib))

    def timestep(self, dt: float):
        r = self.lib.bepuhost_timestep(self.h, dt)
        if r == -1:
            raise ValueError(_err(self.lib))  # ArgumentException (Simulation.cs:318-319)
        if r != 0:
            raise RuntimeError(_err(self.lib))


def diff_type_batch(batch: int, type_id: int, bodies: int, prestep_floats: int, old_handles, old_references, new_handles, new_references_aosoa, new_prestep_aosoa):
    """The C++ twin of integration/csharp/HipTimestepper.cs's DiffTypeBatch (host/bepu_host.cpp): the operations (int32 [n, 8], bepuhip_structural_op) and payload words that
    turn a device type batch holding `old_handles` / `old_references` ([count, bodies], encoded) into this frame's arrangement."""
    lib = load_library()
    old_h = np.ascontiguousarray(old_handles, dtype=np.int32)
    old_r = np.ascontiguousarray(old_references, dtype=np.int32).reshape(-1)
    new_h = np.ascontiguousarray(new_handles, dtype=np.int32)
    new_r = np.ascontiguousarray(new_references_aosoa, dtype=np.int32)
    new_p = np.ascontiguousarray(new_prestep_aosoa, dtype=np.float32)
    capacity = 4 * (old_h.size + new_h.size) + 16
    ops = np.zeros((capacity, 8), dtype=np.int32)
    payload = np.zeros(max(1, new_h.size * (bodies + prestep_floats)), dtype=np.uint32)
    words = C.c_int32()
    n = lib.bepuhost_diff_type_batch(batch, type_id, bodies, prestep_floats, old_h.ctypes.data, old_h.size, old_r.ctypes.data, new_h.ctypes.data, new_h.size, new_r.ctypes.data,
                                     new_p.ctypes.data, ops.ctypes.data, capacity, payload.ctypes.data, payload.size, C.byref(words))
    if n < 0:
        raise RuntimeError("diff_type_batch: capacity")
    return ops[:n].copy(), payload[: max(1, words.value)].copy()


def diff_type_batch_identities(batch: int, type_id: int, bodies: int, prestep_floats: int, old_handles, old_references, old_body_handles, new_handles, new_references_aosoa,
                               new_body_handles, new_prestep_aosoa):
    """diff_type_batch with the constraints' identities (constraint handle + body handles): also returns, per constraint of the new arrangement, its index in the old one
    (-1: new). A constraint handle that names another constraint now (Solver.HandlePool reuses handles) comes out as a removal plus an addition."""
    lib = load_library()
    old_h = np.ascontiguousarray(old_handles, dtype=np.int32)
    old_r = np.ascontiguousarray(old_references, dtype=np.int32).reshape(-1)
    old_b = np.ascontiguousarray(old_body_handles, dtype=np.int32).reshape(-1)
    new_h = np.ascontiguousarray(new_handles, dtype=np.int32)
    new_r = np.ascontiguousarray(new_references_aosoa, dtype=np.int32)
    new_b = np.ascontiguousarray(new_body_handles, dtype=np.int32).reshape(-1)
    new_p = np.ascontiguousarray(new_prestep_aosoa, dtype=np.float32)
    capacity = 4 * (old_h.size + new_h.size) + 16
    ops = np.zeros((capacity, 8), dtype=np.int32)
    payload = np.zeros(max(1, new_h.size * (bodies + prestep_floats)), dtype=np.uint32)
    survivor = np.full(max(1, new_h.size), -1, dtype=np.int32)
    words = C.c_int32()
    n = lib.bepuhost_diff_type_batch_identities(batch, type_id, bodies, prestep_floats, old_h.ctypes.data, old_h.size, old_r.ctypes.data, old_b.ctypes.data, new_h.ctypes.data, new_h.size,
                                                new_r.ctypes.data, new_b.ctypes.data, new_p.ctypes.data, ops.ctypes.data, capacity, payload.ctypes.data, payload.size, C.byref(words),
                                                survivor.ctypes.data)
    if n < 0:
        raise RuntimeError("diff_type_batch: capacity")
    return ops[:n].copy(), payload[: max(1, words.value)].copy(), survivor[: new_h.size].copy()
