"""A small mutable mirror of the reference's Solver storage for tests of structural updates: Solver.Add (greedy first-fit batch, BepuPhysics/Solver.cs:1182-1199),
TypeProcessor.AllocateInTypeBatch (append, TypeProcessor.cs:314-334) and TypeProcessor.Remove (swap-with-last, :695-717), kept in per-lane Python lists and
exported as a `Scene` (AOSOA) whenever the oracle or a fresh upload needs one. TEST INFRASTRUCTURE."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from bepuphysics2_amd.scene import BUNDLE_WIDTH, KINEMATIC_MASK, TYPE_TABLE, Scene, TypeBatchData, bundle_count, to_aosoa


class MutableSolver:
    def __init__(self, bodies: np.ndarray, bundle_width: int = BUNDLE_WIDTH):
        self.bodies = np.ascontiguousarray(bodies, dtype=np.float32).copy()
        self.w = bundle_width
        self.batches: List[Dict[int, dict]] = []          # per batch: type_id -> {"refs": [...], "prestep": [...], "acc": [...]}
        self.type_order: List[List[int]] = []             # type batches of a batch in creation order (ConstraintBatch.GetOrCreateTypeBatch)
        self.batch_handles: List[Dict[int, int]] = []     # batchReferencedHandles: dynamic body -> 1
        self.kinematic_constrained: List[int] = []       # Solver.ConstrainedKinematicHandles (Solver.cs:68): kinematic bodies with at least one constraint
        self.kinematic_uses: Dict[int, int] = {}
        self.next_handle = 0                               # constraint handles (Solver.HandlePool): stable for a constraint's life, TypeBatch.IndexToHandle lists them by index

    def is_kinematic(self, body: int) -> bool:
        return not np.any(self.bodies[body, 16:23])

    def add(self, type_id: int, bodies: Sequence[int], prestep_lane: Sequence[float]) -> Tuple[int, int, List[int]]:
        """Returns (batch index, index in type batch, encoded references)."""
        nb, pf, imf, _ = TYPE_TABLE[type_id]
        assert len(bodies) == nb and len(prestep_lane) == pf
        encoded, blocking = [], []
        for b in bodies:
            if self.is_kinematic(b):
                encoded.append(int(b) | KINEMATIC_MASK)
                self.kinematic_uses[int(b)] = self.kinematic_uses.get(int(b), 0) + 1
                if self.kinematic_uses[int(b)] == 1:
                    self.kinematic_constrained.append(int(b))
            else:
                encoded.append(int(b))
                blocking.append(int(b))
        for bi in range(len(self.batches) + 1):
            if bi == len(self.batches):
                self.batches.append({})
                self.type_order.append([])
                self.batch_handles.append({})
            if any(h in self.batch_handles[bi] for h in blocking):
                continue
            tb = self.batches[bi].get(type_id)
            if tb is None:
                tb = self.batches[bi][type_id] = {"refs": [], "prestep": [], "acc": [], "handles": []}
                self.type_order[bi].append(type_id)
            tb["refs"].append(encoded)
            tb["prestep"].append(np.asarray(prestep_lane, dtype=np.float32).copy())
            tb["acc"].append(np.zeros(imf, dtype=np.float32))
            tb["handles"].append(self.next_handle)
            self.next_handle += 1
            for h in blocking:
                self.batch_handles[bi][h] = 1
            return bi, len(tb["refs"]) - 1, encoded
        raise AssertionError("unreachable")

    def remove(self, batch_index: int, type_id: int, index: int):
        tb = self.batches[batch_index][type_id]
        for r in tb["refs"][index]:
            if not (r & KINEMATIC_MASK):
                del self.batch_handles[batch_index][r]
            else:  # RemoveConstraintReferencesFromBodiesEnumerator (Solver.cs:1368-1377): a kinematic body's last constraint takes it out of ConstrainedKinematicHandles (FastRemove)
                body = int(r) & ~KINEMATIC_MASK
                self.kinematic_uses[body] -= 1
                if self.kinematic_uses[body] == 0:
                    at = self.kinematic_constrained.index(body)
                    self.kinematic_constrained[at] = self.kinematic_constrained[-1]
                    self.kinematic_constrained.pop()
        last = len(tb["refs"]) - 1
        if index < last:
            for key in ("refs", "prestep", "acc", "handles"):
                tb[key][index] = tb[key][last]
        for key in ("refs", "prestep", "acc", "handles"):
            tb[key].pop()

    def swap(self, batch_index: int, type_id: int, a: int, b: int):
        """Two constraints of a type batch change places (bepuhip_swap_constraints; the reference has no such call)."""
        tb = self.batches[batch_index][type_id]
        for key in ("refs", "prestep", "acc", "handles"):
            tb[key][a], tb[key][b] = tb[key][b], tb[key][a]

    def snapshot(self) -> Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray]]:
        """What a diffing host keeps from one frame to the next: per (batch, type id) the handles by index and the encoded references by index."""
        out = {}
        for bi, b in enumerate(self.batches):
            for t in self.type_order[bi]:
                d = b[t]
                nb = TYPE_TABLE[t][0]
                out[(bi, t)] = (np.asarray(d["handles"], dtype=np.int32).copy(), np.asarray(d["refs"], dtype=np.int32).reshape(len(d["refs"]), nb).copy())
        return out

    def remove_body(self, index: int) -> List[Tuple[int, int, int, int, int]]:
        """Bodies.RemoveAt (BepuPhysics/BodySet.cs:83-110) of a body without constraints: the last body takes its slot, and every constraint that referenced the last
        body is patched (Solver.UpdateForBodyMemoryMove, Solver.cs:1475 -> TypeProcessor.UpdateForBodyMemoryMove, TypeProcessor.cs:807). Returns the patches as
        (batch index, type id, index in type batch, body slot, new encoded reference) — what bepuhip_update_body_reference is called with."""
        last = self.bodies.shape[0] - 1
        for b in self.batches:
            for tb in b.values():
                assert all((int(r) & ~KINEMATIC_MASK) != index for refs in tb["refs"] for r in refs), "the removed body still has constraints"
        patches = []
        if index != last:
            for bi, b in enumerate(self.batches):
                for t in self.type_order[bi]:
                    for i, refs in enumerate(b[t]["refs"]):
                        for k, r in enumerate(refs):
                            if (int(r) & ~KINEMATIC_MASK) == last:
                                refs[k] = index | (int(r) & KINEMATIC_MASK)
                                patches.append((bi, t, i, k, int(refs[k])))
                if last in self.batch_handles[bi]:
                    self.batch_handles[bi][index] = self.batch_handles[bi].pop(last)
            if self.kinematic_uses.get(last, 0) > 0:
                self.kinematic_uses[index] = self.kinematic_uses.pop(last)
                self.kinematic_constrained[self.kinematic_constrained.index(last)] = index
            self.bodies[index] = self.bodies[last]
        self.kinematic_uses.pop(last, None)
        self.bodies = np.ascontiguousarray(self.bodies[:last])
        return patches

    def dynamic_degree(self, body: int) -> int:
        """Constraints that reference the (dynamic) body: a dynamic body appears at most once per batch."""
        return sum(1 for handles in self.batch_handles if body in handles)

    def locations(self, predicate=lambda type_id: True) -> List[Tuple[int, int, int]]:
        return [(bi, t, i) for bi, b in enumerate(self.batches) for t in self.type_order[bi] if predicate(t) for i in range(len(b[t]["refs"]))]

    def to_scene(self) -> Scene:
        batches = []
        for bi, b in enumerate(self.batches):
            tbs = []
            for t in self.type_order[bi]:
                d = b[t]
                n = len(d["refs"])
                nb, pf, imf, _ = TYPE_TABLE[t]
                refs = np.asarray(d["refs"], dtype=np.int32).reshape(n, nb)
                pre = np.asarray(d["prestep"], dtype=np.float32).reshape(n, pf)
                acc = np.asarray(d["acc"], dtype=np.float32).reshape(n, imf)
                tbs.append(TypeBatchData(t, n, to_aosoa(refs, self.w, fill=-1) if n else np.zeros(0, np.int32), to_aosoa(pre, self.w) if n else np.zeros(0, np.float32),
                                         to_aosoa(acc, self.w) if n else np.zeros(0, np.float32)))
            batches.append(tbs)
        ident = np.arange(self.bodies.shape[0], dtype=np.int32)
        return Scene(self.bodies.copy(), ident.copy(), ident.copy(), batches, np.asarray(self.kinematic_constrained, dtype=np.int32), self.w)

    def absorb(self, scene: Scene):
        """Take bodies, accumulated impulses and prestep (contact depths) back from a solved export of this solver."""
        self.bodies[...] = scene.bodies
        for bi, tbs in enumerate(scene.batches):
            for tb in tbs:
                d = self.batches[bi][tb.type_id]
                acc, pre = tb.accumulated_lanes(self.w), tb.prestep_lanes(self.w)
                for i in range(tb.count):
                    d["acc"][i] = acc[i].copy()
                    d["prestep"][i] = pre[i].copy()
