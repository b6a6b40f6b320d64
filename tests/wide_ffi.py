"""ctypes wrapper of oracle/wide/libbepu_wide.so — TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

oracle/wide is the second, independently transcribed CPU restatement (AOSOA-8 SIMD, the reference's own shape). It takes the same
marshalled scene as oracle_ffi (the structs are a data format) and solves IN PLACE."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import oracle_ffi
from oracle_ffi import OracleParams, OracleScene, _Marshalled, _p

WIDE_DIR = os.path.join(oracle_ffi.ORACLE_DIR, "wide")
_libs = {}


def load(variant: str = "") -> C.CDLL:
    """variant: "" (-O2 checker), "fast" (-O3, bench.py's cpu_baseline), "zerominus" (Vector<T> unary minus as Zero - v, wide_vec.h),
    "rcpx86" (MathHelper.FastReciprocal[SquareRoot] as vrcpps / vrsqrtps: the reference's branch on an AVX host, wide_joints_more.h)."""
    name = f"libbepu_wide_{variant}.so" if variant else "libbepu_wide.so"
    if name not in _libs:
        path = os.path.join(WIDE_DIR, name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s"], cwd=oracle_ffi.ORACLE_DIR)
        lib = C.CDLL(path)
        lib.wide_solve.argtypes = [C.POINTER(OracleScene), C.POINTER(OracleParams)]
        lib.wide_solve.restype = C.c_int
        lib.wide_session_create.argtypes = [C.POINTER(OracleScene), C.POINTER(OracleParams), C.POINTER(C.c_int)]
        lib.wide_session_create.restype = C.c_void_p
        lib.wide_session_solve.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_double)]
        lib.wide_session_solve.restype = C.c_int
        lib.wide_session_read.argtypes = [C.c_void_p]
        lib.wide_session_read.restype = C.c_int
        lib.wide_session_destroy.argtypes = [C.c_void_p]
        lib.wide_session_destroy.restype = None
        lib.wide_pin_plan.argtypes = [C.POINTER(C.c_int)]
        lib.wide_pin_plan.restype = None
        lib.wide_math_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.wide_constraint_iterate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        lib.wide_constraint_iterate.restype = C.c_int
        _libs[name] = lib
    return _libs[name]


def pin_plan(variant: str = "") -> dict:
    """Where oracle/wide puts its workers (wide_solver.cpp, PinPlan): CPUs the process may use, physical cores / hardware threads of the socket the workers fill first."""
    out = (C.c_int * 4)()
    load(variant).wide_pin_plan(out)
    return {"cpus_allowed": int(out[0]), "first_socket_physical_cores": int(out[1]), "first_socket_hardware_threads": int(out[2]), "pinned": bool(out[3])}


def _params(dt, solve_description, callbacks, threads):
    its = np.ascontiguousarray(solve_description.iterations(), dtype=np.int32)
    p = OracleParams()
    p.dt = float(dt)
    p.substep_count = int(solve_description.substep_count)
    p.velocity_iterations = _p(its)
    p.gravity[0], p.gravity[1], p.gravity[2] = [float(x) for x in callbacks.gravity]
    p.linear_damping = float(callbacks.linear_damping)
    p.angular_damping = float(callbacks.angular_damping)
    p.allow_substeps_for_unconstrained = int(bool(callbacks.allow_substeps_for_unconstrained_bodies))
    p.integrate_velocity_for_kinematics = int(bool(callbacks.integrate_velocity_for_kinematics))
    p.threads = int(threads)
    p.angular_integration_mode = int(getattr(callbacks, "angular_integration_mode", 0))
    p.fallback_batch_threshold = int(solve_description.fallback_batch_threshold)
    return p, (its, oracle_ffi.apply_velocity_model(p, callbacks))  # what the parameters point at stays alive with the second value


def solve(scene, dt, solve_description, callbacks, threads: int = 1, fast: bool = False, variant: str = ""):
    """Simulation.Solve through oracle/wide, IN PLACE on ``scene``'s buffers (bundle width must be 8)."""
    lib = load("fast" if fast else variant)
    assert scene.bodies.flags["C_CONTIGUOUS"] and scene.bodies.dtype == np.float32
    m = _Marshalled(scene)
    p, _its = _params(dt, solve_description, callbacks, threads)
    rc = lib.wide_solve(C.byref(m.c), C.byref(p))
    if rc != 0:
        raise RuntimeError(f"wide_solve failed: {rc}")


class Session:
    """The scene held as the reference holds it between frames (oracle/wide/wide_solver.cpp `Session`): aligned buffers owned by the library, batch handle sets built once.
    ``solve(frames, threads)`` runs Simulation.Solve's three calls and nothing else; ``read()`` copies the state back into ``scene``'s buffers."""

    def __init__(self, scene, dt, solve_description, callbacks, fast: bool = False, variant: str = ""):
        self.lib = load("fast" if fast else variant)
        assert scene.bodies.flags["C_CONTIGUOUS"] and scene.bodies.dtype == np.float32
        self._m = _Marshalled(scene)  # keeps the caller's buffers alive: read() writes into them
        self.dt = float(dt)
        p, self._its = _params(dt, solve_description, callbacks, 1)
        status = C.c_int(0)
        self.handle = self.lib.wide_session_create(C.byref(self._m.c), C.byref(p), C.byref(status))
        if not self.handle:
            raise RuntimeError(f"wide_session_create failed: {status.value}")

    def solve(self, frames: int = 1, threads: int = 1):
        """Returns the seconds spent in (PrepareConstraintIntegrationResponsibilities, Solve, IntegrateAfterSubstepping) over the frames, and the seconds all
        workers together spent inside Solve's work blocks."""
        phases = (C.c_double * 4)()
        rc = self.lib.wide_session_solve(self.handle, self.dt, int(threads), int(frames), phases)
        if rc != 0:
            raise RuntimeError(f"wide_session_solve failed: {rc}")
        return tuple(phases)

    def read(self):
        self.lib.wide_session_read(self.handle)

    def close(self):
        if self.handle:
            self.lib.wide_session_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def math_probe(x):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, c, a = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    lib.wide_math_probe(_p(x), x.size, _p(s), _p(c), _p(a))
    return s, c, a


def constraint_iterate(type_id, body_a, body_b, prestep, accumulated, dt, iterations):
    rc = load().wide_constraint_iterate(type_id, _p(body_a), _p(body_b), _p(prestep), _p(accumulated), float(dt), int(iterations))
    if rc != 0:
        raise RuntimeError(f"wide_constraint_iterate failed: {rc}")


def predict_bounding_boxes(bodies, dt, callbacks, collidables, hulls=None, compounds=None, meshes=None):
    """PoseIntegrator.PredictBoundingBoxes through oracle/wide's BoundingBoxBatcher transcription (wide_bounds.h); same records as oracle_ffi.predict_bounding_boxes, bundle width 8."""
    from bepuphysics2_amd.native import COLLIDABLE_DTYPE, COMPOUND_CHILD_DTYPE, PREDICTED_BOUNDS_DTYPE
    lib = load()
    hulls, compounds, meshes = hulls or [], compounds or [], meshes or []
    b = np.ascontiguousarray(bodies, dtype=np.float32)
    c = np.ascontiguousarray(collidables, dtype=COLLIDABLE_DTYPE)
    out = np.zeros(c.shape[0], dtype=PREDICTED_BOUNDS_DTYPE)
    p = OracleParams()
    p.dt = float(dt)
    p.substep_count = 1
    p.gravity[0], p.gravity[1], p.gravity[2] = [float(x) for x in callbacks.gravity]
    p.linear_damping = float(callbacks.linear_damping)
    p.angular_damping = float(callbacks.angular_damping)
    p.integrate_velocity_for_kinematics = int(bool(callbacks.integrate_velocity_for_kinematics))
    _gravity_table = oracle_ffi.apply_velocity_model(p, callbacks)  # noqa: F841
    pts = np.ascontiguousarray(np.concatenate([np.asarray(h, dtype=np.float32).reshape(-1, 3) for h in hulls]) if hulls else np.zeros((0, 3), np.float32), dtype=np.float32)
    hull_begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(h) for h in hulls])]), dtype=np.int32)
    kids = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=COMPOUND_CHILD_DTYPE).reshape(-1) for k in compounds]) if compounds else np.zeros(0, COMPOUND_CHILD_DTYPE),
                                dtype=COMPOUND_CHILD_DTYPE)
    kid_begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(k) for k in compounds])]), dtype=np.int32)
    tris = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float32).reshape(-1, 9) for t, _ in meshes]) if meshes else np.zeros((0, 9), np.float32), dtype=np.float32)
    tri_begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([np.asarray(t).reshape(-1, 9).shape[0] for t, _ in meshes])]), dtype=np.int32)
    scales = np.ascontiguousarray(np.asarray([s for _, s in meshes], dtype=np.float32).reshape(-1, 3))
    fn = lib.wide_predict_bounding_boxes
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(OracleParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int]
    fn.restype = C.c_int
    rc = fn(_p(b), c.shape[0], C.byref(p), _p(c), _p(out), _p(pts), hull_begin.ctypes.data_as(C.c_void_p), len(hulls), _p(kids), kid_begin.ctypes.data_as(C.c_void_p), len(compounds), _p(tris),
            tri_begin.ctypes.data_as(C.c_void_p), _p(scales), len(meshes))
    if rc != 0:
        raise RuntimeError(f"wide_predict_bounding_boxes failed: {rc}")
    return out
