"""ctypes wrapper of oracle/libbepu_oracle.so — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")


class OracleTypeBatch(C.Structure):
    _fields_ = [("type_id", C.c_int32), ("constraint_count", C.c_int32), ("body_refs", C.c_void_p), ("prestep", C.c_void_p), ("accumulated", C.c_void_p)]


class OracleParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("substep_count", C.c_int32), ("velocity_iterations", C.c_void_p), ("gravity", C.c_float * 3),
                ("linear_damping", C.c_float), ("angular_damping", C.c_float), ("allow_substeps_for_unconstrained", C.c_int32),
                ("integrate_velocity_for_kinematics", C.c_int32), ("threads", C.c_int32), ("exchange", C.c_void_p), ("exchange_user", C.c_void_p), ("angular_integration_mode", C.c_int32),
                ("fallback_batch_threshold", C.c_int32),  # 0 = SolveDescription.DefaultFallbackBatchThreshold (64)
                ("velocity_model", C.c_int32), ("planet_center", C.c_float * 3), ("planet_gravity", C.c_float), ("body_gravity", C.c_void_p)]  # which IntegrateVelocity (scene.PoseIntegratorCallbacks)


def apply_velocity_model(p: "OracleParams", callbacks):
    """The velocity model of ``callbacks`` into the oracle's parameters; returns what must stay alive while the parameters are in use."""
    p.velocity_model = int(getattr(callbacks, "velocity_model", 0))
    centre = getattr(callbacks, "planet_center", (0.0, 0.0, 0.0))
    p.planet_center[0], p.planet_center[1], p.planet_center[2] = [float(x) for x in centre]
    p.planet_gravity = float(getattr(callbacks, "planet_gravity", 0.0))
    table = getattr(callbacks, "body_gravity", None)
    keep = None
    if table is not None:
        keep = np.ascontiguousarray(table, dtype=np.float32)
        p.body_gravity = keep.ctypes.data
    else:
        p.body_gravity = None
    return keep


class OracleScene(C.Structure):
    _fields_ = [("bodies", C.c_void_p), ("body_count", C.c_int32), ("index_to_handle", C.c_void_p), ("handle_to_index", C.c_void_p),
                ("handle_capacity", C.c_int32), ("batch_count", C.c_int32), ("type_batch_counts", C.c_void_p), ("type_batches", C.c_void_p),
                ("constrained_kinematic_handles", C.c_void_p), ("constrained_kinematic_count", C.c_int32), ("bundle_width", C.c_int32)]


_libs = {}


def load(fast: bool = False) -> C.CDLL:
    name = "libbepu_oracle_fast.so" if fast else "libbepu_oracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s"], cwd=ORACLE_DIR)
    lib = C.CDLL(path)
    lib.oracle_solve.argtypes = [C.POINTER(OracleScene), C.POINTER(OracleParams)]
    lib.oracle_solve.restype = C.c_int
    lib.oracle_prepare_flags.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.oracle_prepare_flags.restype = C.c_int
    lib.oracle_constraint_iterate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int]
    lib.oracle_constraint_iterate.restype = C.c_int
    lib.oracle_math_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_type_info.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 4
    lib.oracle_predict_bounding_boxes.argtypes = [C.c_void_p, C.c_int, C.POINTER(OracleParams), C.c_void_p, C.c_void_p]
    lib.oracle_type_info.restype = C.c_int
    _libs[name] = lib
    return lib


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


class _Marshalled:
    """Keeps numpy buffers alive while the C structs point into them."""

    def __init__(self, scene):
        self.scene = scene
        tbs = [tb for b in scene.batches for tb in b]
        self.tb_array = (OracleTypeBatch * max(len(tbs), 1))()
        for i, tb in enumerate(tbs):
            self.tb_array[i] = OracleTypeBatch(tb.type_id, tb.count, _p(tb.body_refs), _p(tb.prestep), _p(tb.accumulated))
        self.counts = np.asarray([len(b) for b in scene.batches], dtype=np.int32)
        self.kin = np.ascontiguousarray(scene.constrained_kinematic_handles, dtype=np.int32)
        self.i2h = np.ascontiguousarray(scene.index_to_handle, dtype=np.int32)
        self.h2i = np.ascontiguousarray(scene.handle_to_index, dtype=np.int32)
        s = OracleScene()
        s.bodies = _p(scene.bodies)
        s.body_count = scene.body_count
        s.index_to_handle = _p(self.i2h)
        s.handle_to_index = _p(self.h2i)
        s.handle_capacity = int(self.h2i.size)
        s.batch_count = len(scene.batches)
        s.type_batch_counts = _p(self.counts)
        s.type_batches = C.cast(self.tb_array, C.c_void_p)
        s.constrained_kinematic_handles = _p(self.kin)
        s.constrained_kinematic_count = int(self.kin.size)
        s.bundle_width = scene.bundle_width
        self.c = s


EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32)


def solve(scene, dt, solve_description, callbacks, threads: int = 1, fast: bool = False, exchange=None):
    """Run the oracle's Simulation.Solve restatement IN PLACE on ``scene``'s buffers."""
    lib = load(fast)
    assert scene.bodies.flags["C_CONTIGUOUS"] and scene.bodies.dtype == np.float32
    m = _Marshalled(scene)
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
    _gravity_table = apply_velocity_model(p, callbacks)  # noqa: F841  (kept alive until the call returns)
    failure = []
    fn = None
    if exchange is not None:  # exchange(substep, pass) after every pass: the CPU stand-in of HipSolver.solve_exchanged
        def trampoline(_user, substep, pass_index):
            try:
                exchange(int(substep), int(pass_index))
                return 0
            except Exception as e:  # noqa: BLE001
                failure.append(e)
                return 1
        fn = EXCHANGE_FN(trampoline)
        p.exchange = C.cast(fn, C.c_void_p)
    rc = lib.oracle_solve(C.byref(m.c), C.byref(p))
    if failure:
        raise failure[0]
    if rc != 0:
        raise RuntimeError(f"oracle_solve failed: {rc}")


def predict_bounding_boxes(bodies, dt, callbacks, collidables, hulls=None, compounds=None, meshes=None, bundle_width=8):
    """PoseIntegrator.PredictBoundingBoxes restated (oracle/bepu_bounds.h): returns PREDICTED_BOUNDS_DTYPE records, bodies untouched."""
    from bepuphysics2_amd.native import COLLIDABLE_DTYPE, PREDICTED_BOUNDS_DTYPE
    lib = load()
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
    _gravity_table = apply_velocity_model(p, callbacks)  # noqa: F841
    if compounds or meshes or bundle_width != 8:
        from bepuphysics2_amd.native import COMPOUND_CHILD_DTYPE
        hulls, compounds, meshes = hulls or [], compounds or [], meshes or []
        pts = np.ascontiguousarray(np.concatenate([np.asarray(h, dtype=np.float32).reshape(-1, 3) for h in hulls]) if hulls else np.zeros((0, 3), np.float32), dtype=np.float32)
        hull_begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(h) for h in hulls])]), dtype=np.int32)
        kids = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=COMPOUND_CHILD_DTYPE).reshape(-1) for k in compounds]) if compounds else np.zeros(0, COMPOUND_CHILD_DTYPE),
                                    dtype=COMPOUND_CHILD_DTYPE)
        kid_begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(k) for k in compounds])]), dtype=np.int32)
        tris = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float32).reshape(-1, 9) for t, _ in meshes]) if meshes else np.zeros((0, 9), np.float32), dtype=np.float32)
        tri_begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([np.asarray(t).reshape(-1, 9).shape[0] for t, _ in meshes])]), dtype=np.int32)
        scales = np.ascontiguousarray(np.asarray([s for _, s in meshes], dtype=np.float32).reshape(-1, 3))
        fn = lib.oracle_predict_bounding_boxes_shapes
        fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(OracleParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                       C.c_void_p, C.c_int, C.c_int]
        fn.restype = C.c_int
        rc = fn(_p(b), c.shape[0], C.byref(p), _p(c), _p(out), _p(pts), hull_begin.ctypes.data_as(C.c_void_p), len(hulls), _p(kids), kid_begin.ctypes.data_as(C.c_void_p), len(compounds),
                _p(tris), tri_begin.ctypes.data_as(C.c_void_p), _p(scales), len(meshes), int(bundle_width))
    elif hulls:
        pts = np.ascontiguousarray(np.concatenate([np.asarray(h, dtype=np.float32).reshape(-1, 3) for h in hulls]), dtype=np.float32)
        begin = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(h) for h in hulls])]), dtype=np.int32)
        lib.oracle_predict_bounding_boxes_hulls.argtypes = [C.c_void_p, C.c_int, C.POINTER(OracleParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.oracle_predict_bounding_boxes_hulls.restype = C.c_int
        rc = lib.oracle_predict_bounding_boxes_hulls(_p(b), c.shape[0], C.byref(p), _p(c), _p(out), _p(pts), _p(begin), len(hulls))
    else:
        rc = lib.oracle_predict_bounding_boxes(_p(b), c.shape[0], C.byref(p), _p(c), _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle_predict_bounding_boxes failed: {rc}")
    return out


def prepare_flags(scene):
    """Returns (merged handle bitset uint64[], flags uint64[] concatenated, coarse uint8[] per flattened type batch)."""
    lib = load()
    m = _Marshalled(scene)
    words = (int(m.h2i.size) + 63) // 64 or 1
    merged = np.zeros(words, dtype=np.uint64)
    cap = sum(((tb.count + 63) // 64) * tb.bodies for b in scene.batches[1:] for tb in b) + 1
    flags = np.zeros(cap, dtype=np.uint64)
    coarse = np.zeros(max(sum(len(b) for b in scene.batches), 1), dtype=np.uint8)
    rc = lib.oracle_prepare_flags(C.byref(m.c), _p(merged), _p(flags), cap, _p(coarse))
    if rc != 0:
        raise RuntimeError(f"oracle_prepare_flags failed: {rc}")
    return merged, flags[:cap - 1], coarse


def constraint_iterate(type_id, body_a, body_b, prestep, accumulated, dt, iterations):
    lib = load()
    rc = lib.oracle_constraint_iterate(type_id, _p(body_a), _p(body_b), _p(prestep), _p(accumulated), float(dt), int(iterations))
    if rc != 0:
        raise RuntimeError(f"oracle_constraint_iterate failed: {rc}")


def math_probe(x):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, c, a = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    lib.oracle_math_probe(_p(x), x.size, _p(s), _p(c), _p(a))
    return s, c, a
