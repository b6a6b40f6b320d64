"""ctypes binding of the C ABI in include/bepuhip.h (libbepuhip.so) and the host-side mirror of the reference's
``Simulation.Solve`` call sequence (BepuPhysics/Simulation.cs:278-290) on top of it.

There is no CPU fallback: if the HIP extension or a GPU is missing, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .scene import PoseIntegratorCallbacks, Scene, SolveDescription, TYPE_TABLE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BEPUHIP_LIB") or os.path.join(_HERE, "csrc", "libbepuhip.so")  # override: A/B timing of kernel builds (tools/)

BEPUHIP_OK = 0
BEPUHIP_E_INVALID_ARGUMENT = -1
BEPUHIP_E_UNSUPPORTED = -2
BEPUHIP_E_DEVICE = -3
BEPUHIP_E_STATE = -4
BEPUHIP_FLAG_NO_GRAPH = 1
BEPUHIP_FLAG_NO_CLUSTERS = 2
BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS = 8  # island schedule: spare device slots per cluster segment, so that add_constraint keeps the island schedule
BEPUHIP_FLAG_EXCLUSIVE_DEVICE = 16  # nothing else runs on the device during a solve: split-island plans launch plainly instead of cooperatively

# Every symbol include/bepuhip.h declares (checked by the CPU test-suite against the header).
EXPORTED_SYMBOLS = [
    "bepuhip_last_error", "bepuhip_create", "bepuhip_destroy", "bepuhip_set_bodies", "bepuhip_begin_constraints",
    "bepuhip_set_type_batch", "bepuhip_end_constraints", "bepuhip_set_constrained_kinematics", "bepuhip_solve",
    "bepuhip_get_bodies", "bepuhip_get_accumulated_impulses", "bepuhip_get_prestep", "bepuhip_get_constrained_flags",
    "bepuhip_last_solve_ms", "bepuhip_set_solve_timing", "bepuhip_set_profiling", "bepuhip_get_profile", "bepuhip_last_constraint_iterations",
    "bepuhip_get_stream", "bepuhip_solve_async", "bepuhip_sync", "bepuhip_reset_state", "bepuhip_type_info",
    "bepuhip_set_cluster_trace", "bepuhip_get_cluster_trace", "bepuhip_get_cluster_cycles", "bepuhip_get_row_policy", "bepuhip_debug_status",
    "bepuhip_set_boundary_bodies", "bepuhip_boundary_deltas", "bepuhip_boundary_apply", "bepuhip_solve_exchanged",
    "bepuhip_colour_constraints", "bepuhip_set_exchange_mode", "bepuhip_set_boundary_layout", "bepuhip_comm_unique_id", "bepuhip_comm_init", "bepuhip_comm_adopt", "bepuhip_solve_lattice",
    "bepuhip_update_bodies", "bepuhip_update_prestep", "bepuhip_update_accumulated_impulses",
    "bepuhip_get_bodies_range", "bepuhip_get_prestep_range", "bepuhip_get_accumulated_impulses_range",
    "bepuhip_predict_bounding_boxes", "bepuhip_set_collidables", "bepuhip_set_convex_hulls", "bepuhip_set_compounds", "bepuhip_set_meshes",
    "bepuhip_set_velocity_model", "bepuhip_solve_with_substep_events", "bepuhip_add_constraint", "bepuhip_remove_constraint", "bepuhip_update_body_reference", "bepuhip_swap_constraints", "bepuhip_apply_structural_ops", "bepuhip_get_constraint_count", "bepuhip_get_schedule", "bepuhip_replan",
    "bepuhip_register_host_memory", "bepuhip_unregister_host_memory", "bepuhip_get_poses_and_velocities", "bepuhip_get_poses_and_velocities_async", "bepuhip_update_prestep_async", "bepuhip_update_accumulated_impulses_async",
    "bepuhip_transfer_rows_async",
    "bepuhip_set_device_group", "bepuhip_get_shared_records", "bepuhip_set_peer_records", "bepuhip_export_shared_records", "bepuhip_import_peer_records",
    "bepuhip_get_owned_bodies", "bepuhip_get_owned_constraints", "bepuhip_sync_owned_bodies",
    "bepuhip_get_kernel_family", "bepuhip_add_constraint_at",
    "bepuhip_replan_begin", "bepuhip_replan_poll", "bepuhip_replan_commit", "bepuhip_replan_cancel",
    "bepuhip_specialise_units", "bepuhip_prebuild_unit",
]


class BepuHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"bepuhip error {code}: {message}")
        self.code = code


class UnsupportedError(BepuHipError):
    """BEPUHIP_E_UNSUPPORTED: the caller should fall back to the reference's own simulation.Solve."""


class Config(C.Structure):
    _fields_ = [("device_ordinal", C.c_int32), ("bundle_width", C.c_int32), ("flags", C.c_int32)]


class VelocityModel(C.Structure):  # bepuhip_velocity_model
    _fields_ = [("model", C.c_int32), ("center", C.c_float * 3), ("gravity", C.c_float)]


class RowTransfer(C.Structure):  # bepuhip_row_transfer
    _fields_ = [("kind", C.c_int32), ("batch_index", C.c_int32), ("type_id", C.c_int32), ("first_bundle", C.c_int32), ("bundle_count", C.c_int32), ("reserved", C.c_int32),
                ("bundles", C.c_void_p)]


ROWS_UPDATE_PRESTEP, ROWS_UPDATE_IMPULSES, ROWS_GET_PRESTEP, ROWS_GET_IMPULSES = 0, 1, 2, 3


class Integrator(C.Structure):
    _fields_ = [("gravity", C.c_float * 3), ("linear_damping", C.c_float), ("angular_damping", C.c_float),
                ("angular_integration_mode", C.c_int32), ("allow_substeps_for_unconstrained", C.c_int32),
                ("integrate_velocity_for_kinematics", C.c_int32)]


EXCHANGE_PER_PASS_AVERAGE, EXCHANGE_PER_BATCH_EXACT = 0, 1
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32)  # bepuhip_exchange_fn(user, substep, pass)
SUBSTEP_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)  # bepuhip_substep_fn(user, substep index)

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """Load libbepuhip.so (built in-tree by __graft_entry__.build()). Raises if it is missing — no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.bepuhip_last_error.restype = C.c_char_p
    lib.bepuhip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.bepuhip_destroy.argtypes = [vp]
    lib.bepuhip_set_bodies.argtypes = [vp, vp, i32]
    lib.bepuhip_begin_constraints.argtypes = [vp, i32, i32]
    lib.bepuhip_set_type_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.bepuhip_end_constraints.argtypes = [vp]
    lib.bepuhip_set_constrained_kinematics.argtypes = [vp, vp, i32]
    lib.bepuhip_solve.argtypes = [vp, f32, i32, vp, C.POINTER(Integrator)]
    lib.bepuhip_solve_async.argtypes = [vp, f32, i32, vp, C.POINTER(Integrator)]
    lib.bepuhip_sync.argtypes = [vp]
    lib.bepuhip_get_bodies.argtypes = [vp, vp, i32]
    lib.bepuhip_get_accumulated_impulses.argtypes = [vp, i32, i32, vp]
    lib.bepuhip_get_prestep.argtypes = [vp, i32, i32, vp]
    lib.bepuhip_get_constrained_flags.argtypes = [vp, vp, i32]
    lib.bepuhip_last_solve_ms.argtypes = [vp, C.POINTER(f32)]
    lib.bepuhip_set_solve_timing.argtypes = [vp, i32]
    lib.bepuhip_set_profiling.argtypes = [vp, i32]
    lib.bepuhip_get_profile.argtypes = [vp, i32, C.POINTER(f32), C.POINTER(i32)]
    lib.bepuhip_last_constraint_iterations.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.bepuhip_get_stream.argtypes = [vp, C.POINTER(vp)]
    lib.bepuhip_reset_state.argtypes = [vp]
    lib.bepuhip_set_cluster_trace.argtypes = [vp, i32]
    lib.bepuhip_set_boundary_bodies.argtypes = [vp, vp, i32]
    lib.bepuhip_boundary_deltas.argtypes = [vp, vp, i32]
    lib.bepuhip_boundary_apply.argtypes = [vp, vp, i32]
    lib.bepuhip_set_exchange_mode.argtypes = [vp, i32]
    lib.bepuhip_colour_constraints.argtypes = [i32, vp, i32, i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    lib.bepuhip_set_boundary_layout.argtypes = [vp, vp, i32, vp]
    lib.bepuhip_comm_unique_id.argtypes = [vp]
    lib.bepuhip_comm_init.argtypes = [vp, vp, i32, i32]
    lib.bepuhip_comm_adopt.argtypes = [vp, vp, i32]
    lib.bepuhip_solve_exchanged.argtypes = [vp, f32, i32, vp, C.POINTER(Integrator), EXCHANGE_FN, vp]
    lib.bepuhip_solve_lattice.argtypes = [vp, f32, i32, vp, C.POINTER(Integrator)]
    lib.bepuhip_get_cluster_cycles.argtypes = [vp, vp, i32, C.POINTER(i32)]
    lib.bepuhip_get_row_policy.argtypes = [vp, C.POINTER(i32)]
    lib.bepuhip_debug_status.argtypes = [vp, vp]
    lib.bepuhip_get_cluster_trace.argtypes = [vp, vp, C.c_int64, C.POINTER(i32)]
    lib.bepuhip_type_info.argtypes = [i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.bepuhip_predict_bounding_boxes.argtypes = [vp, f32, C.POINTER(Integrator), vp, i32, vp]
    lib.bepuhip_set_collidables.argtypes = [vp, vp, i32]
    lib.bepuhip_set_convex_hulls.argtypes = [vp, vp, vp, i32]
    lib.bepuhip_set_compounds.argtypes = [vp, vp, vp, i32]
    lib.bepuhip_set_meshes.argtypes = [vp, vp, vp, vp, i32]
    lib.bepuhip_update_bodies.argtypes = [vp, vp, i32, i32]
    lib.bepuhip_get_bodies_range.argtypes = [vp, vp, i32, i32]
    lib.bepuhip_register_host_memory.argtypes = [vp, vp, C.c_int64]
    lib.bepuhip_unregister_host_memory.argtypes = [vp, vp]
    lib.bepuhip_get_poses_and_velocities.argtypes = [vp, vp, i32]
    lib.bepuhip_get_poses_and_velocities_async.argtypes = [vp, vp, i32]
    for name in ("bepuhip_update_prestep", "bepuhip_update_prestep_async", "bepuhip_update_accumulated_impulses", "bepuhip_update_accumulated_impulses_async", "bepuhip_get_prestep_range", "bepuhip_get_accumulated_impulses_range"):
        getattr(lib, name).argtypes = [vp, i32, i32, i32, i32, vp]
    lib.bepuhip_transfer_rows_async.argtypes = [vp, vp, i32]
    lib.bepuhip_set_device_group.argtypes = [vp, i32, i32]
    lib.bepuhip_get_shared_records.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.bepuhip_set_peer_records.argtypes = [vp, i32, vp]
    lib.bepuhip_export_shared_records.argtypes = [vp, vp]
    lib.bepuhip_import_peer_records.argtypes = [vp, i32, vp]
    lib.bepuhip_get_owned_bodies.argtypes = [vp, vp, i32]
    lib.bepuhip_get_owned_constraints.argtypes = [vp, i32, i32, vp]
    lib.bepuhip_sync_owned_bodies.argtypes = [vp]
    lib.bepuhip_add_constraint.argtypes = [vp, i32, i32, vp, vp, C.POINTER(i32)]
    lib.bepuhip_add_constraint_at.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.bepuhip_remove_constraint.argtypes = [vp, i32, i32, i32]
    lib.bepuhip_update_body_reference.argtypes = [vp, i32, i32, i32, i32, i32]
    lib.bepuhip_set_velocity_model.argtypes = [vp, C.POINTER(VelocityModel), vp, i32]
    lib.bepuhip_solve_with_substep_events.argtypes = [vp, f32, i32, vp, C.POINTER(Integrator), SUBSTEP_FN, SUBSTEP_FN, vp]
    lib.bepuhip_swap_constraints.argtypes = [vp, i32, i32, i32, i32]
    lib.bepuhip_apply_structural_ops.argtypes = [vp, vp, i32, vp, i32, C.POINTER(i32)]
    lib.bepuhip_get_constraint_count.argtypes = [vp, i32, i32, C.POINTER(i32)]
    lib.bepuhip_get_schedule.argtypes = [vp, C.POINTER(i32)]
    lib.bepuhip_get_kernel_family.argtypes = [vp, C.POINTER(i32)]
    lib.bepuhip_replan.argtypes = [vp]
    lib.bepuhip_replan_begin.argtypes = [vp]
    lib.bepuhip_replan_poll.argtypes = [vp, C.POINTER(i32)]
    lib.bepuhip_replan_commit.argtypes = [vp, i32, C.POINTER(i32)]
    lib.bepuhip_replan_cancel.argtypes = [vp]
    lib.bepuhip_specialise_units.argtypes = [vp, i32, C.POINTER(i32)]
    lib.bepuhip_prebuild_unit.argtypes = [C.c_uint64, i32, i32, C.c_char_p, i32]
    for name in EXPORTED_SYMBOLS:
        if name != "bepuhip_last_error":
            getattr(lib, name).restype = i32
    _lib = lib
    return lib


def _check(lib, status: int):
    if status == BEPUHIP_OK:
        return
    msg = lib.bepuhip_last_error().decode("utf-8", "replace")
    if status == BEPUHIP_E_UNSUPPORTED:
        raise UnsupportedError(status, msg)
    if status == BEPUHIP_E_INVALID_ARGUMENT:
        raise ValueError(msg)  # the reference throws ArgumentException here (Simulation.cs:318-319, SolveDescription.cs:42-47)
    raise BepuHipError(status, msg)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def make_integrator(cb: PoseIntegratorCallbacks) -> Integrator:
    integ = Integrator()
    integ.gravity[0], integ.gravity[1], integ.gravity[2] = [float(x) for x in cb.gravity]
    integ.linear_damping = float(cb.linear_damping)
    integ.angular_damping = float(cb.angular_damping)
    integ.angular_integration_mode = int(getattr(cb, "angular_integration_mode", 0))
    integ.allow_substeps_for_unconstrained = int(bool(cb.allow_substeps_for_unconstrained_bodies))
    integ.integrate_velocity_for_kinematics = int(bool(cb.integrate_velocity_for_kinematics))
    return integ


# bepuhip_collidable / bepuhip_predicted_bounds (include/bepuhip.h) as numpy record types
COLLIDABLE_DTYPE = np.dtype([("shape_type", "<i4"), ("shape", "<f4", (9,)), ("minimum_speculative_margin", "<f4"), ("maximum_speculative_margin", "<f4"),
                             ("allow_expansion_beyond_speculative_margin", "<i4"), ("sleep_threshold", "<f4"), ("minimum_timesteps_under_threshold", "<i4"),
                             ("activity", "<i4")])
PREDICTED_BOUNDS_DTYPE = np.dtype([("min", "<f4", (3,)), ("speculative_margin", "<f4"), ("max", "<f4", (3,)), ("activity", "<i4")])
SHAPE_SPHERE, SHAPE_CAPSULE, SHAPE_BOX, SHAPE_TRIANGLE, SHAPE_CYLINDER = 0, 1, 2, 3, 4  # Sphere.Id ... Cylinder.Id
SHAPE_CONVEX_HULL, SHAPE_COMPOUND, SHAPE_BIG_COMPOUND, SHAPE_MESH = 5, 6, 7, 8     # ConvexHull.Id ... Mesh.Id
# bepuhip_compound_child: a convex child shape and its pose in the compound's frame (CompoundChild, BepuPhysics/Collidables/Compound.cs:13-40)
COMPOUND_CHILD_DTYPE = np.dtype([("shape_type", "<i4"), ("shape", "<f4", (9,)), ("local_position", "<f4", (3,)), ("local_orientation", "<f4", (4,))])


class HipSolver:
    """Device mirror of one Simulation's solver state; ``solve`` replaces ``Simulation.Solve`` (Simulation.cs:278-290)."""

    PROFILE_FAMILIES = ("incremental", "integrate", "warmstart", "solve", "final", "cluster")

    def __init__(self, device: int = 0, bundle_width: int = 8, use_graph: bool = True, use_clusters: bool = True, reserve_update_slots: bool = False,
                 exclusive_device: bool = False):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        cfg = Config(device, bundle_width, (0 if use_graph else BEPUHIP_FLAG_NO_GRAPH) | (0 if use_clusters else BEPUHIP_FLAG_NO_CLUSTERS) | (BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS if reserve_update_slots else 0)
                     | (BEPUHIP_FLAG_EXCLUSIVE_DEVICE if exclusive_device else 0))
        _check(self.lib, self.lib.bepuhip_create(C.byref(cfg), C.byref(self.ctx)))
        self.bundle_width = bundle_width
        self._scene_meta = None
        self._boundary_count = 0
        self._exchange_mode = 0

    def close(self):
        if self.ctx:
            self.lib.bepuhip_destroy(self.ctx)  # unregisters what is still registered ...
            self.ctx = C.c_void_p()
        self._registered_arrays = {}            # ... only then may the buffers go

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- upload ----
    def set_bodies(self, bodies: np.ndarray):
        bodies = np.ascontiguousarray(bodies, dtype=np.float32)
        assert bodies.ndim == 2 and bodies.shape[1] == 32
        _check(self.lib, self.lib.bepuhip_set_bodies(self.ctx, _ptr(bodies), bodies.shape[0]))

    def set_constraints(self, scene: Scene, fallback_batch_threshold: int = 64):
        _check(self.lib, self.lib.bepuhip_begin_constraints(self.ctx, len(scene.batches), fallback_batch_threshold))
        for bi, batch in enumerate(scene.batches):
            for tb in batch:
                _check(self.lib, self.lib.bepuhip_set_type_batch(self.ctx, bi, tb.type_id, tb.count, _ptr(tb.body_refs), _ptr(tb.prestep), _ptr(tb.accumulated)))
        _check(self.lib, self.lib.bepuhip_end_constraints(self.ctx))

    def upload(self, scene: Scene, fallback_batch_threshold: int = 64):
        assert scene.bundle_width == self.bundle_width
        self.set_bodies(scene.bodies)
        self.set_constraints(scene, fallback_batch_threshold)
        kin = np.ascontiguousarray(scene.constrained_kinematic_indices(), dtype=np.int32)
        _check(self.lib, self.lib.bepuhip_set_constrained_kinematics(self.ctx, _ptr(kin), kin.size))
        self._scene_meta = [(bi, tb.type_id, tb.count) for bi, b in enumerate(scene.batches) for tb in b]

    # ---- solve ----
    def set_velocity_model(self, callbacks: PoseIntegratorCallbacks):
        """IntegrateVelocity as data (bepuhip_set_velocity_model): sent when the callbacks' model differs from what the context holds."""
        table = getattr(callbacks, "body_gravity", None)
        key = (int(getattr(callbacks, "velocity_model", 0)), tuple(float(x) for x in getattr(callbacks, "planet_center", (0, 0, 0))), float(getattr(callbacks, "planet_gravity", 0.0)),
               None if table is None else np.ascontiguousarray(table, dtype=np.float32).tobytes())
        if key == getattr(self, "_velocity_model_key", (0, (0.0, 0.0, 0.0), 0.0, None)):
            return
        m = VelocityModel()
        m.model = key[0]
        m.center[0], m.center[1], m.center[2] = key[1]
        m.gravity = key[2]
        values = None if table is None else np.ascontiguousarray(table, dtype=np.float32)
        _check(self.lib, self.lib.bepuhip_set_velocity_model(self.ctx, C.byref(m), None if values is None else _ptr(values), 0 if values is None else values.size))
        self._velocity_model_key = key

    def solve(self, dt: float, solve_description: SolveDescription, callbacks: PoseIntegratorCallbacks, asynchronous: bool = False):
        self.set_velocity_model(callbacks)
        its = np.ascontiguousarray(solve_description.iterations(), dtype=np.int32)
        integ = make_integrator(callbacks)
        fn = self.lib.bepuhip_solve_async if asynchronous else self.lib.bepuhip_solve
        _check(self.lib, fn(self.ctx, float(dt), int(solve_description.substep_count), _ptr(its), C.byref(integ)))

    def solve_with_substep_events(self, dt: float, solve_description: SolveDescription, callbacks: PoseIntegratorCallbacks, started=None, ended=None):
        """Solver.SubstepStarted / SubstepEnded (Solver.cs:125-146): ``started(substep)`` / ``ended(substep)`` run around every substep with the stream drained; they may
        use the update_* and get_* calls. Exceptions raised by a handler surface after the solve."""
        self.set_velocity_model(callbacks)
        its = np.ascontiguousarray(solve_description.iterations(), dtype=np.int32)
        integ = make_integrator(callbacks)
        failure = []

        def wrap(fn):
            def trampoline(_user, substep):
                try:
                    if fn is not None and not failure:
                        fn(int(substep))
                except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
                    failure.append(e)
            return SUBSTEP_FN(trampoline)

        a, b = wrap(started), wrap(ended)
        _check(self.lib, self.lib.bepuhip_solve_with_substep_events(self.ctx, float(dt), int(solve_description.substep_count), _ptr(its), C.byref(integ), a, b, None))
        if failure:
            raise failure[0]

    # ---- one connected scene split across ranks (include/bepuhip.h, "solve_exchanged") ----
    def set_boundary_bodies(self, local_indices: np.ndarray):
        idx = np.ascontiguousarray(local_indices, dtype=np.int32)
        self._boundary_count = int(idx.size)
        _check(self.lib, self.lib.bepuhip_set_boundary_bodies(self.ctx, _ptr(idx), idx.size))

    def boundary_deltas(self) -> np.ndarray:
        """[boundary bodies, 6] float32 deltas — or, in the per-batch exact mode, uint32 XOR patterns (same buffer)."""
        out = np.zeros((self._boundary_count, 6), dtype=np.float32)
        _check(self.lib, self.lib.bepuhip_boundary_deltas(self.ctx, _ptr(out), 0))
        return out.view(np.uint32) if self._exchange_mode == EXCHANGE_PER_BATCH_EXACT else out

    def boundary_apply(self, sums: np.ndarray):
        sums = np.ascontiguousarray(sums)
        sums = sums.view(np.float32) if sums.dtype in (np.uint32, np.int32) else sums.astype(np.float32, copy=False)  # XOR patterns travel as they are
        assert sums.shape == (self._boundary_count, 6)
        _check(self.lib, self.lib.bepuhip_boundary_apply(self.ctx, _ptr(sums), 0))

    def boundary_deltas_device(self, device_pointer: int):
        _check(self.lib, self.lib.bepuhip_boundary_deltas(self.ctx, C.c_void_p(device_pointer), 1))

    def boundary_apply_device(self, device_pointer: int):
        _check(self.lib, self.lib.bepuhip_boundary_apply(self.ctx, C.c_void_p(device_pointer), 1))

    def set_exchange_mode(self, mode: int):
        _check(self.lib, self.lib.bepuhip_set_exchange_mode(self.ctx, int(mode)))
        self._exchange_mode = int(mode)

    def set_boundary_layout(self, dense_rows: np.ndarray, dense_row_count: int, holders: Optional[np.ndarray] = None):
        rows = np.ascontiguousarray(dense_rows, dtype=np.int32)
        assert rows.size == self._boundary_count
        h = None if holders is None else np.ascontiguousarray(holders, dtype=np.float32)
        assert h is None or h.size == dense_row_count
        _check(self.lib, self.lib.bepuhip_set_boundary_layout(self.ctx, _ptr(rows), int(dense_row_count), _ptr(h)))

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(self.lib, self.lib.bepuhip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        _check(self.lib, self.lib.bepuhip_comm_init(self.ctx, C.c_char_p(unique_id), int(rank), int(world)))

    def solve_lattice(self, dt: float, solve_description: SolveDescription, callbacks: PoseIntegratorCallbacks):
        """The exchanged frame with the exchange enqueued on the solver's stream (RCCL all-reduce when a communicator is set): no host synchronisation inside."""
        its = np.ascontiguousarray(solve_description.iterations(), dtype=np.int32)
        integ = make_integrator(callbacks)
        _check(self.lib, self.lib.bepuhip_solve_lattice(self.ctx, float(dt), int(solve_description.substep_count), _ptr(its), C.byref(integ)))

    def solve_exchanged(self, dt: float, solve_description: SolveDescription, callbacks: PoseIntegratorCallbacks, exchange):
        """``exchange(substep, pass)`` runs after every pass (0 = warm start, k = k-th velocity iteration) — after every batch in the per-batch exact mode, with
        pass | (batch launch + 1) << 16; exceptions abort the solve."""
        its = np.ascontiguousarray(solve_description.iterations(), dtype=np.int32)
        integ = make_integrator(callbacks)
        failure = []

        def trampoline(_user, substep, pass_index):
            try:
                exchange(int(substep), int(pass_index))
                return 0
            except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
                failure.append(e)
                return 1

        fn = EXCHANGE_FN(trampoline)
        status = self.lib.bepuhip_solve_exchanged(self.ctx, float(dt), int(solve_description.substep_count), _ptr(its), C.byref(integ), fn, None)
        if failure:
            raise failure[0]
        _check(self.lib, status)

    def sync(self):
        _check(self.lib, self.lib.bepuhip_sync(self.ctx))

    def reset_state(self):
        _check(self.lib, self.lib.bepuhip_reset_state(self.ctx))

    # ---- download ----
    def get_bodies(self, count: int) -> np.ndarray:
        out = np.empty((count, 32), dtype=np.float32)
        _check(self.lib, self.lib.bepuhip_get_bodies(self.ctx, _ptr(out), count))
        return out

    def register_host_memory(self, array: np.ndarray):
        """Pins ``array``'s buffer for the life of the context (BufferPool blocks are pinned, BufferPool.cs:42,83): copies from / to it become asynchronous DMA."""
        assert array.flags["C_CONTIGUOUS"]
        _check(self.lib, self.lib.bepuhip_register_host_memory(self.ctx, _ptr(array), array.nbytes))
        # the buffer must outlive its registration: freed memory that is still registered leaves the runtime with a stale pinned range, and whatever the allocator
        # puts there next (a later copy's staging vector, say) is then copied "as pinned memory" — found in round 4 as an invalid-argument hipMemcpy two tests later
        if not hasattr(self, "_registered_arrays"):
            self._registered_arrays = {}
        self._registered_arrays[array.ctypes.data] = array

    def unregister_host_memory(self, array: np.ndarray):
        _check(self.lib, self.lib.bepuhip_unregister_host_memory(self.ctx, _ptr(array)))
        getattr(self, "_registered_arrays", {}).pop(array.ctypes.data, None)

    def get_poses_and_velocities(self, bodies: np.ndarray, asynchronous: bool = False):
        """Writes the MotionState half (floats 0-15) of every BodyDynamics into ``bodies`` ((n, 32) float32, in place); the inertia half stays what it was."""
        assert bodies.flags["C_CONTIGUOUS"] and bodies.dtype == np.float32 and bodies.ndim == 2 and bodies.shape[1] == 32
        fn = self.lib.bepuhip_get_poses_and_velocities_async if asynchronous else self.lib.bepuhip_get_poses_and_velocities
        _check(self.lib, fn(self.ctx, _ptr(bodies), bodies.shape[0]))

    def download(self, scene: Scene):
        """Write device state back into ``scene``'s buffers (bodies, accumulated impulses, prestep)."""
        scene.bodies[...] = self.get_bodies(scene.body_count)
        for bi, batch in enumerate(scene.batches):
            for tb in batch:
                if tb.count == 0:
                    continue
                _check(self.lib, self.lib.bepuhip_get_accumulated_impulses(self.ctx, bi, tb.type_id, _ptr(tb.accumulated)))
                _check(self.lib, self.lib.bepuhip_get_prestep(self.ctx, bi, tb.type_id, _ptr(tb.prestep)))

    # ---- ranged in-place updates / read-backs (what the narrow phase and user code rewrite between frames) ----
    def update_bodies(self, first: int, bodies: np.ndarray):
        b = np.ascontiguousarray(bodies, dtype=np.float32).reshape(-1, 32)
        _check(self.lib, self.lib.bepuhip_update_bodies(self.ctx, _ptr(b), first, b.shape[0]))

    def get_bodies_range(self, first: int, count: int) -> np.ndarray:
        out = np.empty((count, 32), dtype=np.float32)
        _check(self.lib, self.lib.bepuhip_get_bodies_range(self.ctx, _ptr(out), first, count))
        return out

    def _bundle_floats(self, type_id: int, prestep: bool) -> int:
        return (TYPE_TABLE[type_id][1] if prestep else TYPE_TABLE[type_id][2]) * self.bundle_width

    # ---- structural updates: the host's TypeProcessor mutations mirrored on the device rows ----
    def add_constraint(self, batch_index: int, type_id: int, encoded_body_references, prestep_lane) -> int:
        """TypeProcessor.AllocateInTypeBatch (TypeProcessor.cs:314-334): append at index ConstraintCount; returns that index."""
        refs = np.ascontiguousarray(encoded_body_references, dtype=np.int32)
        lane = np.ascontiguousarray(prestep_lane, dtype=np.float32)
        assert refs.size == TYPE_TABLE[type_id][0] and lane.size == TYPE_TABLE[type_id][1]
        index = C.c_int32()
        _check(self.lib, self.lib.bepuhip_add_constraint(self.ctx, batch_index, type_id, _ptr(refs), _ptr(lane), C.byref(index)))
        return int(index.value)

    def add_constraint_at(self, batch_index: int, type_id: int, index: int, encoded_body_references, prestep_lane):
        """An addition to the sequential fallback batch at the lane the reference's allocation chose (bepuhip_add_constraint_at)."""
        refs = np.ascontiguousarray(encoded_body_references, dtype=np.int32)
        lane = np.ascontiguousarray(prestep_lane, dtype=np.float32)
        assert refs.size == TYPE_TABLE[type_id][0] and lane.size == TYPE_TABLE[type_id][1]
        _check(self.lib, self.lib.bepuhip_add_constraint_at(self.ctx, int(batch_index), int(type_id), int(index), _ptr(refs), _ptr(lane)))

    def remove_constraint(self, batch_index: int, type_id: int, index: int):
        """TypeProcessor.Remove, non-fallback (TypeProcessor.cs:695-717): swap-with-last."""
        _check(self.lib, self.lib.bepuhip_remove_constraint(self.ctx, batch_index, type_id, index))

    def swap_constraints(self, batch_index: int, type_id: int, index_a: int, index_b: int):
        """The constraints at two indices of a type batch change places (bepuhip.h: what a host that diffs the reference's type batches needs besides append and swap-with-last)."""
        _check(self.lib, self.lib.bepuhip_swap_constraints(self.ctx, batch_index, type_id, index_a, index_b))

    def apply_structural_ops(self, ops) -> None:
        """One call for a frame's structural changes, in order (bepuhip_apply_structural_ops). `ops`: tuples
        ("add", batch, type_id, encoded_body_references, prestep_lane[, expected_index]) / ("remove", batch, type_id, index) /
        ("update", batch, type_id, index, body_index_in_constraint, encoded_body_reference) / ("swap", batch, type_id, index_a, index_b)."""
        table = np.zeros((len(ops), 8), dtype=np.int32)
        payload = []
        words = 0
        for i, op in enumerate(ops):
            kind = op[0]
            if kind == "add":
                refs = np.ascontiguousarray(op[3], dtype=np.int32)
                lane = np.ascontiguousarray(op[4], dtype=np.float32)
                assert refs.size == TYPE_TABLE[op[2]][0] and lane.size == TYPE_TABLE[op[2]][1]
                table[i] = (0, op[1], op[2], op[5] if len(op) > 5 else -1, 0, 0, words, 0)
                payload += [refs.view(np.uint32), lane.view(np.uint32)]
                words += refs.size + lane.size
            elif kind == "remove":
                table[i] = (1, op[1], op[2], op[3], 0, 0, 0, 0)
            elif kind == "update":
                table[i] = (2, op[1], op[2], op[3], op[4], int(op[5]), 0, 0)
            elif kind == "swap":
                table[i] = (3, op[1], op[2], op[3], op[4], 0, 0, 0)
            else:
                raise ValueError(kind)
        flat = np.concatenate(payload) if payload else np.zeros(1, dtype=np.uint32)
        failed = C.c_int32(-1)
        _check(self.lib, self.lib.bepuhip_apply_structural_ops(self.ctx, _ptr(table), len(ops), _ptr(flat), int(words), C.byref(failed)))

    def apply_structural_op_table(self, table: np.ndarray, payload: np.ndarray) -> None:
        """The same with the operation table (int32 [n, 8]: kind, batch, type id, index, slot, reference, payload offset, 0) and the payload words prepared by the caller."""
        table = np.ascontiguousarray(table, dtype=np.int32)
        payload = np.ascontiguousarray(payload, dtype=np.uint32)
        failed = C.c_int32(-1)
        _check(self.lib, self.lib.bepuhip_apply_structural_ops(self.ctx, _ptr(table), table.shape[0], _ptr(payload), payload.size, C.byref(failed)))

    def update_body_reference(self, batch_index: int, type_id: int, index: int, body_index_in_constraint: int, encoded_body_reference: int):
        """TypeProcessor.UpdateForBodyMemoryMove (TypeProcessor.cs:807)."""
        _check(self.lib, self.lib.bepuhip_update_body_reference(self.ctx, batch_index, type_id, index, body_index_in_constraint, int(encoded_body_reference)))

    def constraint_count(self, batch_index: int, type_id: int) -> int:
        n = C.c_int32()
        _check(self.lib, self.lib.bepuhip_get_constraint_count(self.ctx, batch_index, type_id, C.byref(n)))
        return int(n.value)

    def set_constrained_kinematics(self, indices: np.ndarray):
        """Solver.ConstrainedKinematicHandles as body indices: re-sent whenever structural updates change it (Solver.cs:1025, 1368-1377)."""
        kin = np.ascontiguousarray(indices, dtype=np.int32)
        _check(self.lib, self.lib.bepuhip_set_constrained_kinematics(self.ctx, _ptr(kin), kin.size))

    def schedule(self) -> int:
        """0 launch-per-batch, 1 island-per-workgroup (whole islands), 2 island-per-workgroup on a split-island plan."""
        n = C.c_int32()
        _check(self.lib, self.lib.bepuhip_get_schedule(self.ctx, C.byref(n)))
        return int(n.value)

    def kernel_family(self) -> int:
        """The type-set family of the last island launch: 0 contacts only, 1 the sixteen hot-path types, 2 all 44, 3 a unit compiled for the context's exact types; -1 none yet (bepuhip_get_kernel_family)."""
        n = C.c_int32()
        _check(self.lib, self.lib.bepuhip_get_kernel_family(self.ctx, C.byref(n)))
        return int(n.value)

    def specialise_units(self, wait: bool = False) -> int:
        """bepuhip_specialise_units: the island kernel compiled for exactly this context's constraint types (found in the unit cache, or compiled by hipcc on a thread of the
        library). Returns 0 unavailable, 1 compiling, 2 loaded, 3 the compiler failed."""
        state = C.c_int32(0)
        _check(self.lib, self.lib.bepuhip_specialise_units(self.ctx, 1 if wait else 0, C.byref(state)))
        return int(state.value)

    def replan(self):
        """A fresh plan for the constraints the device holds now (bepuhip_replan): values stay on the device, only the references are read back."""
        _check(self.lib, self.lib.bepuhip_replan(self.ctx))

    def replan_begin(self):
        """bepuhip_replan_begin: the references are read back, a host thread plans them; the frames go on (on the launch-per-batch schedule) until replan_commit."""
        _check(self.lib, self.lib.bepuhip_replan_begin(self.ctx))

    def replan_state(self) -> int:
        """0 no re-plan in flight, 1 planning, 2 ready to commit."""
        state = C.c_int32(0)
        _check(self.lib, self.lib.bepuhip_replan_poll(self.ctx, C.byref(state)))
        return int(state.value)

    def replan_commit(self, wait: bool = False) -> bool:
        """bepuhip_replan_commit: True when the new plan is the context's now (False: still planning and `wait` was not asked for)."""
        committed = C.c_int32(0)
        _check(self.lib, self.lib.bepuhip_replan_commit(self.ctx, 1 if wait else 0, C.byref(committed)))
        return bool(committed.value)

    def replan_cancel(self):
        _check(self.lib, self.lib.bepuhip_replan_cancel(self.ctx))

    def update_prestep(self, batch_index: int, type_id: int, first_bundle: int, bundles: np.ndarray, asynchronous: bool = False):
        """``bundles``: the type batch's PrestepData bundles [first_bundle, first_bundle + n) exactly as the reference stores them (AOSOA).
        ``asynchronous``: enqueued on the context's stream; ``bundles`` must stay unchanged (and alive) until the next ``sync``."""
        b = np.ascontiguousarray(bundles, dtype=np.float32).reshape(-1)
        n, rem = divmod(b.size, self._bundle_floats(type_id, True))
        if rem:
            raise ValueError("prestep data is not a whole number of bundles")
        fn = self.lib.bepuhip_update_prestep_async if asynchronous else self.lib.bepuhip_update_prestep
        _check(self.lib, fn(self.ctx, batch_index, type_id, first_bundle, n, _ptr(b)))

    def update_accumulated_impulses(self, batch_index: int, type_id: int, first_bundle: int, bundles: np.ndarray, asynchronous: bool = False):
        b = np.ascontiguousarray(bundles, dtype=np.float32).reshape(-1)
        n, rem = divmod(b.size, self._bundle_floats(type_id, False))
        if rem:
            raise ValueError("impulse data is not a whole number of bundles")
        fn = self.lib.bepuhip_update_accumulated_impulses_async if asynchronous else self.lib.bepuhip_update_accumulated_impulses
        _check(self.lib, fn(self.ctx, batch_index, type_id, first_bundle, n, _ptr(b)))

    # ---- device groups: one connected scene on several devices, exact (include/bepuhip.h) ----
    def set_device_group(self, world: int, rank: int):
        """Before ``upload``: the plan holds ``world`` devices' worth of clusters and this context runs range ``rank`` of them."""
        _check(self.lib, self.lib.bepuhip_set_device_group(self.ctx, world, rank))

    def shared_records(self):
        """(device pointer, bytes) of this context's copy of the record table; (0, 0) when the plan has no shared bodies."""
        ptr, size = C.c_void_p(), C.c_int64()
        _check(self.lib, self.lib.bepuhip_get_shared_records(self.ctx, C.byref(ptr), C.byref(size)))
        return int(ptr.value or 0), int(size.value)

    def set_peer_records(self, peer: int, pointer: int):
        _check(self.lib, self.lib.bepuhip_set_peer_records(self.ctx, peer, C.c_void_p(pointer)))

    def export_shared_records(self) -> bytes:
        handle = (C.c_ubyte * 64)()
        _check(self.lib, self.lib.bepuhip_export_shared_records(self.ctx, handle))
        return bytes(handle)

    def import_peer_records(self, peer: int, handle: bytes):
        buf = (C.c_ubyte * 64).from_buffer_copy(handle)
        _check(self.lib, self.lib.bepuhip_import_peer_records(self.ctx, peer, buf))

    def owned_bodies(self, count: int) -> np.ndarray:
        mask = np.zeros(count, dtype=np.uint8)
        _check(self.lib, self.lib.bepuhip_get_owned_bodies(self.ctx, _ptr(mask), count))
        return mask.astype(bool)

    def owned_constraints(self, batch_index: int, type_id: int, count: int) -> np.ndarray:
        mask = np.zeros(max(count, 1), dtype=np.uint8)
        _check(self.lib, self.lib.bepuhip_get_owned_constraints(self.ctx, batch_index, type_id, _ptr(mask)))
        return mask[:count].astype(bool)

    def sync_owned_bodies(self):
        _check(self.lib, self.lib.bepuhip_sync_owned_bodies(self.ctx))

    def transfer_rows(self, items):
        """bepuhip_transfer_rows_async: ``items`` = (kind, batch index, type id, first bundle, float32 array of whole bundles) tuples — or a prepared ``RowTransfer`` array from
        ``row_transfer_table`` — enqueued in order on the context's stream. UPDATE kinds read the arrays, GET kinds write them (in place: they must be C-contiguous float32);
        nothing may touch them before the next ``sync``."""
        table = items if isinstance(items, C.Array) else self.row_transfer_table(items)
        _check(self.lib, self.lib.bepuhip_transfer_rows_async(self.ctx, table, len(table)))

    def row_transfer_table(self, items):
        table = (RowTransfer * len(items))()
        for slot, (kind, batch_index, type_id, first_bundle, bundles) in zip(table, items):
            assert bundles.flags["C_CONTIGUOUS"] and bundles.dtype == np.float32
            n, rem = divmod(bundles.size, self._bundle_floats(type_id, kind in (ROWS_UPDATE_PRESTEP, ROWS_GET_PRESTEP)))
            if rem:
                raise ValueError("not a whole number of bundles")
            slot.kind, slot.batch_index, slot.type_id, slot.first_bundle, slot.bundle_count, slot.bundles = kind, batch_index, type_id, first_bundle, n, bundles.ctypes.data
        self._transfer_keepalive = [it[4] for it in items]
        return table

    def get_prestep_range(self, batch_index: int, type_id: int, first_bundle: int, bundle_count: int) -> np.ndarray:
        out = np.empty(bundle_count * self._bundle_floats(type_id, True), dtype=np.float32)
        _check(self.lib, self.lib.bepuhip_get_prestep_range(self.ctx, batch_index, type_id, first_bundle, bundle_count, _ptr(out)))
        return out

    def get_accumulated_impulses_range(self, batch_index: int, type_id: int, first_bundle: int, bundle_count: int) -> np.ndarray:
        out = np.empty(bundle_count * self._bundle_floats(type_id, False), dtype=np.float32)
        _check(self.lib, self.lib.bepuhip_get_accumulated_impulses_range(self.ctx, batch_index, type_id, first_bundle, bundle_count, _ptr(out)))
        return out

    # ---- PredictBoundingBoxes ----
    def set_convex_hulls(self, hulls):
        """hulls: a list of float32 [n_i, 3] point sets; collidables of shape_type 5 name them by index in shape[0]."""
        pts = np.ascontiguousarray(np.concatenate([np.asarray(h, dtype=np.float32).reshape(-1, 3) for h in hulls]) if hulls else np.zeros((0, 3), np.float32), dtype=np.float32)
        begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(h) for h in hulls])]), dtype=np.int32)
        _check(self.lib, self.lib.bepuhip_set_convex_hulls(self.ctx, _ptr(pts), _ptr(begin), len(hulls)))

    def set_compounds(self, compounds):
        """compounds: a list of COMPOUND_CHILD_DTYPE arrays (the children of each Compound / BigCompound); collidables of shape_type 6 / 7 name them by index in shape[0]."""
        kids = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=COMPOUND_CHILD_DTYPE).reshape(-1) for k in compounds]) if compounds else np.zeros(0, COMPOUND_CHILD_DTYPE),
                                    dtype=COMPOUND_CHILD_DTYPE)
        begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(k) for k in compounds])]), dtype=np.int32)
        _check(self.lib, self.lib.bepuhip_set_compounds(self.ctx, _ptr(kids), _ptr(begin), len(compounds)))

    def set_meshes(self, meshes):
        """meshes: a list of (triangles float32 [n_i, 3, 3], scale xyz); collidables of shape_type 8 name them by index in shape[0]."""
        tris = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float32).reshape(-1, 9) for t, _ in meshes]) if meshes else np.zeros((0, 9), np.float32), dtype=np.float32)
        begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([np.asarray(t).reshape(-1, 9).shape[0] for t, _ in meshes])]), dtype=np.int32)
        scales = np.ascontiguousarray(np.asarray([s for _, s in meshes], dtype=np.float32).reshape(-1, 3))
        _check(self.lib, self.lib.bepuhip_set_meshes(self.ctx, _ptr(tris), _ptr(begin), _ptr(scales), len(meshes)))

    def set_collidables(self, collidables: np.ndarray):
        """Keep the collidable records on the device; later ``predict_bounding_boxes(dt, cb)`` calls use (and update the sleep counters of) these."""
        c = np.ascontiguousarray(collidables, dtype=COLLIDABLE_DTYPE)
        _check(self.lib, self.lib.bepuhip_set_collidables(self.ctx, _ptr(c), c.shape[0]))
        self._resident_collidables = c.shape[0]

    def predict_bounding_boxes(self, dt: float, callbacks: PoseIntegratorCallbacks, collidables: Optional[np.ndarray] = None) -> np.ndarray:
        """``collidables``: COLLIDABLE_DTYPE records, one per body index (None: the resident ones). Returns PREDICTED_BOUNDS_DTYPE records (PoseIntegrator.cs:307-370)."""
        self.set_velocity_model(callbacks)
        integ = make_integrator(callbacks)
        if collidables is None:
            n = getattr(self, "_resident_collidables", 0)
            out = np.zeros(n, dtype=PREDICTED_BOUNDS_DTYPE)
            _check(self.lib, self.lib.bepuhip_predict_bounding_boxes(self.ctx, dt, C.byref(integ), None, n, _ptr(out)))
            return out
        c = np.ascontiguousarray(collidables, dtype=COLLIDABLE_DTYPE)
        out = np.zeros(c.shape[0], dtype=PREDICTED_BOUNDS_DTYPE)
        _check(self.lib, self.lib.bepuhip_predict_bounding_boxes(self.ctx, dt, C.byref(integ), _ptr(c), c.shape[0], _ptr(out)))
        return out

    def constrained_flags(self, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=np.uint8)
        _check(self.lib, self.lib.bepuhip_get_constrained_flags(self.ctx, _ptr(out), count))
        return out

    # ---- measurement ----
    def set_solve_timing(self, enabled: bool) -> None:
        """HIP events around every solve from now on (bepuhip_last_solve_ms); off by default: they cost two marker packets per solve."""
        _check(self.lib, self.lib.bepuhip_set_solve_timing(self.ctx, 1 if enabled else 0))

    def last_solve_ms(self) -> float:
        v = C.c_float()
        _check(self.lib, self.lib.bepuhip_last_solve_ms(self.ctx, C.byref(v)))
        return float(v.value)

    def last_constraint_iterations(self) -> int:
        v = C.c_int64()
        _check(self.lib, self.lib.bepuhip_last_constraint_iterations(self.ctx, C.byref(v)))
        return int(v.value)

    def set_profiling(self, enabled: bool):
        _check(self.lib, self.lib.bepuhip_set_profiling(self.ctx, int(enabled)))

    def profile(self) -> dict:
        out = {}
        for i, name in enumerate(self.PROFILE_FAMILIES):
            ms, n = C.c_float(), C.c_int32()
            _check(self.lib, self.lib.bepuhip_get_profile(self.ctx, i, C.byref(ms), C.byref(n)))
            out[name] = (float(ms.value), int(n.value))
        return out

    def row_policy(self) -> int:
        """-1 still measuring, 0 plain, 1 non-temporal constraint-row accesses, 2 plain rows + a span of code touched ahead (island schedule; bepuhip.h)."""
        v = C.c_int32(-1)
        _check(self.lib, self.lib.bepuhip_get_row_policy(self.ctx, C.byref(v)))
        return int(v.value)

    def cluster_cycles(self) -> np.ndarray:
        """Shader clocks per cluster of the last solve (empty when the launch-per-batch schedule ran)."""
        n = C.c_int32()
        _check(self.lib, self.lib.bepuhip_get_cluster_cycles(self.ctx, None, 0, C.byref(n)))
        out = np.zeros(max(int(n.value), 0), dtype=np.uint64)
        if out.size:
            _check(self.lib, self.lib.bepuhip_get_cluster_cycles(self.ctx, _ptr(out), out.size, C.byref(n)))
        return out

    def set_cluster_trace(self, enabled: bool):
        _check(self.lib, self.lib.bepuhip_set_cluster_trace(self.ctx, int(enabled)))

    def cluster_trace(self, passes: int) -> np.ndarray:
        """(passes, items, 8) uint64 records of the first cluster: claim clock, publish clock, wave|type<<8|batch<<16|stage<<32, count, loads landed, before wait, after wait, 0."""
        cap = 1 << 22
        buf = np.zeros(cap, dtype=np.uint64)
        items = C.c_int32()
        _check(self.lib, self.lib.bepuhip_get_cluster_trace(self.ctx, _ptr(buf), cap, C.byref(items)))
        n = int(items.value)
        return buf[: passes * n * 8].reshape(passes, n, 8)

    def stream_handle(self) -> int:
        v = C.c_void_p()
        _check(self.lib, self.lib.bepuhip_get_stream(self.ctx, C.byref(v)))
        return int(v.value or 0)


def type_info(type_id: int):
    lib = load_library()
    b, p, i = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib, lib.bepuhip_type_info(type_id, C.byref(b), C.byref(p), C.byref(i)))
    return b.value, p.value, i.value
