"""Host-side containers in the reference's own memory layouts (numpy views of what the C# BufferPool would hold).

* bodies: ``float32[n, 32]`` — ``Bodies.ActiveSet.DynamicsState``, 128-byte ``BodyDynamics`` (BepuPhysics/BodyProperties.cs:11-46,258-338):
  floats 0-3 orientation xyzw, 4-6 position, 8-10 linear, 12-14 angular, 16-22 local inverse inertia (XX,YX,YY,ZX,ZY,ZZ,invMass),
  24-30 world inverse inertia.
* type batches: AOSOA buffers with bundle width W (BepuUtilities/BundleIndexing.cs:50-60, BepuPhysics/Constraints/TypeProcessor.cs:139-148,269-279).

``SceneBuilder`` is a small pure-Python mirror of ``Bodies.Add`` / ``Solver.Add`` (greedy first-fit batch colouring,
BepuPhysics/Solver.cs:1058-1199) for test-sized scenes; large scenes are built by the C++ host mirror (``bepuphysics2_amd/host``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

KINEMATIC_MASK = 1 << 30  # BepuPhysics/Bodies_GatherScatter.cs:107-118
BODY_REFERENCE_MASK = 0x3FFFFFFF
FALLBACK_BATCH_THRESHOLD = 64  # BepuPhysics/SolveDescription.cs:38
BUNDLE_WIDTH = 8  # Vector<float>.Count on the reference's AVX2 hosts

# type id -> (bodies per constraint, prestep floats per lane, accumulated impulse floats per lane, name)
TYPE_TABLE: Dict[int, tuple] = {
    0: (1, 11, 4, "Contact1OneBody"), 1: (1, 15, 5, "Contact2OneBody"), 2: (1, 19, 6, "Contact3OneBody"), 3: (1, 23, 7, "Contact4OneBody"),
    4: (2, 14, 4, "Contact1"), 5: (2, 18, 5, "Contact2"), 6: (2, 22, 6, "Contact3"), 7: (2, 26, 7, "Contact4"),
    22: (2, 8, 3, "BallSocket"), 23: (2, 8, 2, "AngularHinge"), 25: (2, 9, 1, "SwingLimit"), 26: (2, 14, 1, "TwistServo"),
    27: (2, 12, 1, "TwistLimit"), 30: (2, 5, 3, "AngularMotor"), 31: (2, 9, 6, "Weld"), 46: (2, 14, 4, "SwivelHinge"), 47: (2, 14, 5, "Hinge"),
    24: (2, 8, 1, "AngularSwivelHinge"), 28: (2, 9, 1, "TwistMotor"), 29: (2, 9, 3, "AngularServo"), 33: (2, 12, 1, "DistanceServo"), 34: (2, 10, 1, "DistanceLimit"),
    41: (2, 6, 1, "AngularAxisMotor"), 42: (1, 9, 3, "OneBodyAngularServo"), 43: (1, 5, 3, "OneBodyAngularMotor"), 44: (1, 11, 3, "OneBodyLinearServo"),
    45: (1, 8, 3, "OneBodyLinearMotor"), 52: (2, 8, 3, "BallSocketMotor"), 53: (2, 11, 3, "BallSocketServo"),
    8: (1, 18, 6, "Contact2NonconvexOneBody"), 9: (1, 25, 9, "Contact3NonconvexOneBody"), 10: (1, 32, 12, "Contact4NonconvexOneBody"),
    15: (2, 21, 6, "Contact2Nonconvex"), 16: (2, 28, 9, "Contact3Nonconvex"), 17: (2, 35, 12, "Contact4Nonconvex"),
    37: (2, 14, 2, "PointOnLineServo"), 38: (2, 15, 1, "LinearAxisServo"), 39: (2, 12, 1, "LinearAxisMotor"), 40: (2, 13, 1, "LinearAxisLimit"),
    54: (2, 6, 1, "AngularAxisGearMotor"),
    32: (4, 3, 1, "VolumeConstraint"), 35: (2, 3, 1, "CenterDistanceConstraint"), 36: (3, 3, 1, "AreaConstraint"), 55: (2, 4, 1, "CenterDistanceLimit"),
}
TYPE_IDS_BY_NAME = {v[3]: k for k, v in TYPE_TABLE.items()}
# The sixteen types of SURVEY.md 8(a) rows a7-a13 (the committed tests/golden/small_scenes.npz fixtures were generated from exactly these).
HOT_PATH_TYPES = [0, 1, 2, 3, 4, 5, 6, 7, 22, 23, 25, 26, 27, 30, 46, 47]
# 8(f) widening, one type at a time.
WIDENED_TYPES = [t for t in sorted(TYPE_TABLE) if t not in HOT_PATH_TYPES]


def bundle_count(count: int, w: int = BUNDLE_WIDTH) -> int:
    return (count + w - 1) // w


def aosoa_index(i: np.ndarray, f: int, fields: int, w: int) -> np.ndarray:
    """Float index of field ``f`` of constraint ``i`` (BepuUtilities/BundleIndexing.cs:50-60)."""
    return (i // w) * (fields * w) + f * w + (i % w)


def to_aosoa(lanes: np.ndarray, w: int = BUNDLE_WIDTH, fill=0) -> np.ndarray:
    """[count, fields] -> AOSOA flat buffer of bundle_count*fields*w elements."""
    count, fields = lanes.shape
    nb = max(bundle_count(count, w), 0)
    out = np.full((nb, fields, w), fill, dtype=lanes.dtype)
    if count:
        idx = np.arange(count)
        out[idx // w, :, idx % w] = lanes
    return out.reshape(-1)


def from_aosoa(buf: np.ndarray, count: int, fields: int, w: int = BUNDLE_WIDTH) -> np.ndarray:
    """AOSOA flat buffer -> [count, fields]."""
    nb = bundle_count(count, w)
    v = buf.reshape(nb, fields, w)
    idx = np.arange(count)
    return v[idx // w, :, idx % w].copy()


@dataclass
class TypeBatchData:
    type_id: int
    count: int
    body_refs: np.ndarray  # int32 AOSOA
    prestep: np.ndarray  # float32 AOSOA
    accumulated: np.ndarray  # float32 AOSOA

    @property
    def bodies(self) -> int:
        return TYPE_TABLE[self.type_id][0]

    def occupied(self, w: int = BUNDLE_WIDTH) -> np.ndarray:
        """bool[count]: False for the empty lanes (-1 references) a sequential-fallback type batch leaves inside its bundles (TypeProcessor.cs:451-560)."""
        v = self.body_refs.reshape(bundle_count(self.count, w), self.bodies, w)[:, 0, :].reshape(-1)[:self.count]
        return v != -1

    @property
    def prestep_floats(self) -> int:
        return TYPE_TABLE[self.type_id][1]

    @property
    def impulse_floats(self) -> int:
        return TYPE_TABLE[self.type_id][2]

    def refs_lanes(self, w: int = BUNDLE_WIDTH) -> np.ndarray:
        return from_aosoa(self.body_refs, self.count, self.bodies, w)

    def prestep_lanes(self, w: int = BUNDLE_WIDTH) -> np.ndarray:
        return from_aosoa(self.prestep, self.count, self.prestep_floats, w)

    def accumulated_lanes(self, w: int = BUNDLE_WIDTH) -> np.ndarray:
        return from_aosoa(self.accumulated, self.count, self.impulse_floats, w)

    def copy(self) -> "TypeBatchData":
        return TypeBatchData(self.type_id, self.count, self.body_refs.copy(), self.prestep.copy(), self.accumulated.copy())


@dataclass
class Scene:
    bodies: np.ndarray  # float32 [n, 32]
    index_to_handle: np.ndarray  # int32 [n]
    handle_to_index: np.ndarray  # int32 [handle_capacity]
    batches: List[List[TypeBatchData]]
    constrained_kinematic_handles: np.ndarray  # int32
    bundle_width: int = BUNDLE_WIDTH

    @property
    def body_count(self) -> int:
        return int(self.bodies.shape[0])

    @property
    def constraint_count(self) -> int:
        """Constraints that exist: a type batch of the sequential fallback batch counts its empty lanes in `count` (ConstraintCount = highest index + 1)."""
        return sum(int(tb.occupied(self.bundle_width).sum()) if tb.count else 0 for b in self.batches for tb in b)

    def copy(self) -> "Scene":
        return Scene(self.bodies.copy(), self.index_to_handle.copy(), self.handle_to_index.copy(),
                     [[tb.copy() for tb in b] for b in self.batches], self.constrained_kinematic_handles.copy(), self.bundle_width)

    def constrained_kinematic_indices(self) -> np.ndarray:
        return self.handle_to_index[self.constrained_kinematic_handles].astype(np.int32)

    def summary(self) -> str:
        per_type: Dict[str, int] = {}
        for b in self.batches:
            for tb in b:
                per_type[TYPE_TABLE[tb.type_id][3]] = per_type.get(TYPE_TABLE[tb.type_id][3], 0) + (int(tb.occupied(self.bundle_width).sum()) if tb.count else 0)
        return f"{self.body_count} bodies, {self.constraint_count} constraints, {len(self.batches)} batches, {per_type}"


@dataclass
class SolveDescription:
    """Mirror of BepuPhysics/SolveDescription.cs:16-136. Note the constructor order: (velocityIterationCount, substepCount) (:55)."""
    velocity_iteration_count: int = 1
    substep_count: int = 1
    fallback_batch_threshold: int = FALLBACK_BATCH_THRESHOLD
    velocity_iteration_scheduler: Optional[object] = None  # callable(substep_index) -> int; <1 means use velocity_iteration_count (:33)

    def __post_init__(self):
        if self.substep_count < 1:
            raise ValueError("Substep count must be positive.")  # SolveDescription.cs:42-47 (ArgumentException)
        if self.velocity_iteration_count < 1:
            raise ValueError("Velocity iteration count must be positive.")
        if self.fallback_batch_threshold < 1:
            raise ValueError("Fallback batch threshold must be positive.")

    def iterations(self) -> np.ndarray:
        """GetVelocityIterationCountForSubstepIndex for every substep (BepuPhysics/Solver_Solve.cs:743-751)."""
        out = np.empty(self.substep_count, dtype=np.int32)
        for s in range(self.substep_count):
            n = self.velocity_iteration_count
            if self.velocity_iteration_scheduler is not None:
                scheduled = int(self.velocity_iteration_scheduler(s))
                if scheduled >= 1:
                    n = scheduled
            out[s] = n
        return out


@dataclass
class PoseIntegratorCallbacks:
    """IPoseIntegratorCallbacks as data: the three properties, and IntegrateVelocity as one of the models that cross the C ABI (bepuhip_velocity_model):
    0 DemoPoseIntegratorCallbacks (Demos/DemoCallbacks.cs:20-109: gravity + damping), 1 PerBodyGravityDemoCallbacks (Demos/Demos/PerBodyGravityDemo.cs:57-88:
    ``body_gravity[i]`` added to the linear Y velocity of the body at index i), 2 PlanetaryGravityCallbacks (Demos/Demos/PlanetDemo.cs:36-47)."""
    gravity: Sequence[float] = (0.0, -10.0, 0.0)
    linear_damping: float = 0.03
    angular_damping: float = 0.03
    allow_substeps_for_unconstrained_bodies: bool = False
    integrate_velocity_for_kinematics: bool = False
    angular_integration_mode: int = 0  # AngularIntegrationMode: 0 Nonconserving, 1 ConserveMomentum, 2 ConserveMomentumWithGyroscopicTorque (PoseIntegrator.cs:20-38)
    velocity_model: int = 0
    planet_center: Sequence[float] = (0.0, 0.0, 0.0)
    planet_gravity: float = 0.0
    body_gravity: Optional[np.ndarray] = None


def make_body(position=(0, 0, 0), orientation=(0, 0, 0, 1), linear=(0, 0, 0), angular=(0, 0, 0),
              inverse_inertia=(1, 0, 1, 0, 0, 1), inverse_mass=1.0) -> np.ndarray:
    """One BodyDynamics record. Kinematic = all-zero inverse inertia and mass (BepuPhysics/Bodies.cs:326-349)."""
    b = np.zeros(32, dtype=np.float32)
    b[0:4] = orientation
    b[4:7] = position
    b[8:11] = linear
    b[12:15] = angular
    b[16:22] = inverse_inertia
    b[22] = inverse_mass
    return b


class SceneBuilder:
    """Pure-Python mirror of Bodies.Add + Solver.Add for small scenes (BepuPhysics/Solver.cs:1058-1199)."""

    def __init__(self, bundle_width: int = BUNDLE_WIDTH, fallback_batch_threshold: int = FALLBACK_BATCH_THRESHOLD):
        self.w = bundle_width
        self.fallback_batch_threshold = fallback_batch_threshold
        self._bodies: List[np.ndarray] = []
        self._batches: List[Dict[int, dict]] = []  # per batch: type_id -> {"refs": [], "prestep": []}
        self._batch_type_order: List[List[int]] = []
        self._batch_handles: List[set] = []
        self._kinematic_constrained: List[int] = []

    def add_body(self, body: np.ndarray) -> int:
        self._bodies.append(np.asarray(body, dtype=np.float32).copy())
        return len(self._bodies) - 1  # handle == index (no removals in this mirror)

    def is_kinematic(self, handle: int) -> bool:
        return not np.any(self._bodies[handle][16:23])

    def add_constraint(self, type_id: int, body_handles: Sequence[int], prestep_lane: Sequence[float]) -> tuple:
        nb, pf, _, _ = TYPE_TABLE[type_id]
        assert len(body_handles) == nb and len(prestep_lane) == pf, (type_id, len(body_handles), len(prestep_lane))
        encoded, blocking = [], []
        for h in body_handles:  # GetBlockingBodyHandles, Solver.cs:1058-1078
            if self.is_kinematic(h):
                encoded.append(h | KINEMATIC_MASK)
                if h not in self._kinematic_constrained:
                    self._kinematic_constrained.append(h)
            else:
                encoded.append(h)
                blocking.append(h)
        for bi in range(len(self._batches) + 1):  # Solver.cs:1189-1196
            if bi == len(self._batches):
                self._batches.append({})
                self._batch_type_order.append([])
                self._batch_handles.append(set())
            if bi == self.fallback_batch_threshold:
                return self._add_to_fallback(bi, type_id, encoded, blocking, prestep_lane)
            if any(h in self._batch_handles[bi] for h in blocking):
                continue
            tb = self._batches[bi].get(type_id)
            if tb is None:
                tb = self._batches[bi][type_id] = {"refs": [], "prestep": []}
                self._batch_type_order[bi].append(type_id)
            tb["refs"].append(encoded)
            tb["prestep"].append(list(prestep_lane))
            self._batch_handles[bi].update(blocking)
            return bi, len(tb["refs"]) - 1
        raise AssertionError("unreachable")

    def _add_to_fallback(self, bi, type_id, encoded, blocking, prestep_lane) -> tuple:
        """The sequential fallback batch (batch index == FallbackBatchThreshold, Solver.cs:1878-1884): bodies may repeat across its bundles, never inside one.
        TypeProcessor.AllocateInTypeBatchForFallback (TypeProcessor.cs:451-560): put the constraint into the first probed bundle that has an empty lane and
        references none of its bodies (AllowFallbackBundleAllocation :338-359 compares masked indices, kinematic or not); otherwise open a new bundle. Lanes
        left empty carry -1 references. (The reference probes at most 17 bundles, hashed by handle, once a type batch has more; this mirror probes them all:
        every layout that keeps the per-bundle invariant is a legal input of the solver.)"""
        w = self.w
        nb, pf, _, _ = TYPE_TABLE[type_id]
        tb = self._batches[bi].get(type_id)
        if tb is None:
            tb = self._batches[bi][type_id] = {"refs": [], "prestep": []}
            self._batch_type_order[bi].append(type_id)
        masked = {e & BODY_REFERENCE_MASK for e in encoded}
        refs, pre = tb["refs"], tb["prestep"]
        target = None
        for b0 in range(0, len(refs), w):
            lanes = refs[b0:b0 + w]
            if any((r & BODY_REFERENCE_MASK) in masked for lane in lanes if lane[0] != -1 for r in lane):
                continue
            holes = [b0 + l for l, lane in enumerate(lanes) if lane[0] == -1]
            if holes:
                target = holes[0]
                break
            if len(lanes) < w:
                target = len(refs)
                break
        if target is None:  # a new bundle: its remaining lanes start out empty
            while len(refs) % w:
                refs.append([-1] * nb)
                pre.append([0.0] * pf)
            target = len(refs)
        if target == len(refs):
            refs.append(encoded)
            pre.append(list(prestep_lane))
        else:
            refs[target], pre[target] = encoded, list(prestep_lane)
        self._batch_handles[bi].update(blocking)
        return bi, target

    def build(self) -> Scene:
        n = len(self._bodies)
        bodies = np.stack(self._bodies).astype(np.float32) if n else np.zeros((0, 32), np.float32)
        # Kinematics' world inertia slot is pre-zeroed at add time (BepuPhysics/BodySet.cs:131-134); dynamics' is refreshed by the solver.
        batches: List[List[TypeBatchData]] = []
        for bi, b in enumerate(self._batches):
            tbs = []
            for type_id in self._batch_type_order[bi]:
                d = b[type_id]
                nb, pf, imf, _ = TYPE_TABLE[type_id]
                refs = np.asarray(d["refs"], dtype=np.int32).reshape(-1, nb)
                pre = np.asarray(d["prestep"], dtype=np.float32).reshape(-1, pf)
                cnt = refs.shape[0]
                tbs.append(TypeBatchData(type_id, cnt, to_aosoa(refs, self.w, fill=-1), to_aosoa(pre, self.w),
                                         np.zeros(bundle_count(cnt, self.w) * imf * self.w, dtype=np.float32)))
            batches.append(tbs)
        ident = np.arange(n, dtype=np.int32)
        return Scene(bodies, ident.copy(), ident.copy(), batches, np.asarray(self._kinematic_constrained, dtype=np.int32), self.w)
