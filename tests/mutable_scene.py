"""A small mutable mirror of the reference's Solver storage for tests of structural updates: Solver.Add (greedy first-fit batch, BepuPhysics/Solver.cs:1182-1199),
TypeProcessor.AllocateInTypeBatch (append, TypeProcessor.cs:314-334) and TypeProcessor.Remove (swap-with-last, :695-717), kept in per-lane Python lists and
exported as a `Scene` (AOSOA) whenever the oracle or a fresh upload needs one. TEST INFRASTRUCTURE."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from bepuphysics2_amd.scene import BUNDLE_WIDTH, KINEMATIC_MASK, TYPE_TABLE, Scene, TypeBatchData, bundle_count, to_aosoa


class MutableSolver:
    def __init__(self, bodies: np.ndarray, bundle_width: int = BUNDLE_WIDTH, fallback_batch_threshold: int = 64):
        self.bodies = np.ascontiguousarray(bodies, dtype=np.float32).copy()
        self.w = bundle_width
        self.fallback_batch_threshold = fallback_batch_threshold  # SolveDescription.FallbackBatchThreshold: Batches[threshold] is the sequential fallback batch (Solver.cs:1878-1884)
        self.free_handles: List[int] = []                  # Solver.HandlePool is last-in-first-out (IdPool)
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
            if bi == self.fallback_batch_threshold:
                return self._add_to_fallback(bi, type_id, encoded, blocking, prestep_lane)
            if any(h in self.batch_handles[bi] for h in blocking):
                continue
            tb = self.batches[bi].get(type_id)
            if tb is None:
                tb = self.batches[bi][type_id] = {"refs": [], "prestep": [], "acc": [], "handles": []}
                self.type_order[bi].append(type_id)
            tb["refs"].append(encoded)
            tb["prestep"].append(np.asarray(prestep_lane, dtype=np.float32).copy())
            tb["acc"].append(np.zeros(imf, dtype=np.float32))
            tb["handles"].append(self._take_handle())
            for h in blocking:
                self.batch_handles[bi][h] = 1
            return bi, len(tb["refs"]) - 1, encoded
        raise AssertionError("unreachable")

    # ---- the sequential fallback batch (round 6): the reference's own allocation and removal rules ----
    @staticmethod
    def rehash(value: int) -> int:
        """HashHelper.Rehash (BepuUtilities/Collections/QuickDictionary.cs:20-41) as a 32-bit signed int."""
        u = (value * 982451653) & 0xFFFFFFFF
        rot = lambda x, k: ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF  # noqa: E731
        r = rot(u, 6) ^ rot(u, 13) ^ rot(u, 25)
        return r - (1 << 32) if r & 0x80000000 else r

    def _take_handle(self) -> int:
        """A handle nothing has had before. (The reference's pool is last-in-first-out; this mirror identifies bodies by index, and a diffing host needs the BODY handles to tell
        "a reused constraint handle" from "a constraint whose body moved in memory" — tests/test_structural_diff.py and the C++ twin, which have them, cover the reuse.)"""
        self.next_handle += 1
        return self.next_handle - 1

    def _add_to_fallback(self, bi, type_id, encoded, blocking, prestep_lane):
        """TypeProcessor.AllocateInTypeBatchForFallback (TypeProcessor.cs:451-571): probe bundles for one that holds none of the constraint's bodies (masked indices:
        kinematic references never block, :338-359) and has an empty lane — all of them while the type batch has at most 17 bundles, else the last one and sixteen more
        chosen from the constraint's handle (HashHelper.Rehash) — and take its first empty lane; no such bundle: lane 0 of a new bundle. Empty lanes carry -1 references and a
        handle of -1; ConstraintCount is the highest index + 1."""
        w = self.w
        nb, pf, imf, _ = TYPE_TABLE[type_id]
        tb = self.batches[bi].get(type_id)
        if tb is None:
            tb = self.batches[bi][type_id] = {"refs": [], "prestep": [], "acc": [], "handles": []}
            self.type_order[bi].append(type_id)
        handle = self._take_handle()
        refs = tb["refs"]
        count = len(refs)
        bundles = (count + w - 1) // w
        masked = [e & ~KINEMATIC_MASK for e in encoded]  # (the broadcast indices have the kinematic flag stripped; the bundle's own references keep theirs: a kinematic never matches)

        def probe(b):
            lanes = refs[b * w:(b + 1) * w]
            for lane in lanes:
                if lane[0] != -1 and any(r in masked for r in lane):
                    return None
            for l in range(w):  # the first lane that holds -1 (lanes beyond ConstraintCount in the last bundle are -1 too)
                if b * w + l >= count or lanes[l][0] == -1:
                    return b * w + l
            return None

        target = None
        if bundles <= 17:
            for b in range(bundles):
                target = probe(b)
                if target is not None:
                    break
        else:
            last = bundles - 1
            target = probe(last)
            if target is None:
                nxt = (self.rehash(handle) & 0x7FFFFFFF) % last
                jump = bundles // 16
                remainder = last - jump * 16
                for k in range(16):
                    target = probe(nxt)
                    if target is not None:
                        break
                    nxt += jump
                    if k < remainder:
                        nxt += 1
                    if nxt >= bundles:
                        nxt -= bundles
        if target is None:
            target = bundles * w
        while len(refs) <= target:
            refs.append([-1] * nb)
            tb["prestep"].append(np.zeros(pf, dtype=np.float32))
            tb["acc"].append(np.zeros(imf, dtype=np.float32))
            tb["handles"].append(-1)
        refs[target] = list(encoded)
        tb["prestep"][target] = np.asarray(prestep_lane, dtype=np.float32).copy()
        tb["acc"][target] = np.zeros(imf, dtype=np.float32)
        tb["handles"][target] = handle
        for h in blocking:  # (the fallback batch's referenced handles count constraints per body; here: presence)
            self.batch_handles[bi][h] = self.batch_handles[bi].get(h, 0) + 1
        return bi, target, encoded

    def _remove_from_fallback(self, batch_index: int, type_id: int, index: int):
        """TypeProcessor.Remove with isFallback (TypeProcessor.cs:633-694)."""
        w = self.w
        tb = self.batches[batch_index][type_id]
        nb = TYPE_TABLE[type_id][0]
        assert tb["refs"][index][0] != -1
        self.free_handles.append(tb["handles"][index])
        tb["handles"][index] = -1
        tb["refs"][index] = [-1] * nb
        count = len(tb["refs"])
        bundle = index // w
        if all(lane[0] == -1 for lane in tb["refs"][bundle * w:(bundle + 1) * w]):
            last_bundle = (count + w - 1) // w - 1
            if bundle != last_bundle:  # the last bundle's memory overwrites the dead bundle's, lanes beyond ConstraintCount included (they are empty)
                nb_, pf_, imf_, _ = TYPE_TABLE[type_id]
                pads = {"refs": lambda: [-1] * nb_, "prestep": lambda: np.zeros(pf_, dtype=np.float32), "acc": lambda: np.zeros(imf_, dtype=np.float32), "handles": lambda: -1}
                for key in ("refs", "prestep", "acc", "handles"):
                    src = list(tb[key][last_bundle * w:(last_bundle + 1) * w])
                    while len(src) < w:
                        src.append(pads[key]())
                    tb[key][bundle * w:(bundle + 1) * w] = src
                last_bundle -= 1
            inner = 0
            if last_bundle >= 0:
                for l, lane in enumerate(tb["refs"][last_bundle * w:(last_bundle + 1) * w]):
                    if lane[0] != -1:
                        inner = l + 1
            new_count = max(0, last_bundle * w + inner)
            for key in ("refs", "prestep", "acc", "handles"):
                del tb[key][new_count:]

    def remove(self, batch_index: int, type_id: int, index: int):
        tb = self.batches[batch_index][type_id]
        if batch_index == self.fallback_batch_threshold:
            for r in tb["refs"][index]:
                if not (r & KINEMATIC_MASK):
                    self.batch_handles[batch_index][r] -= 1
                    if self.batch_handles[batch_index][r] == 0:
                        del self.batch_handles[batch_index][r]
                else:
                    body = int(r) & ~KINEMATIC_MASK
                    self.kinematic_uses[body] -= 1
                    if self.kinematic_uses[body] == 0:
                        at = self.kinematic_constrained.index(body)
                        self.kinematic_constrained[at] = self.kinematic_constrained[-1]
                        self.kinematic_constrained.pop()
            self._remove_from_fallback(batch_index, type_id, index)
            return
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
        self.free_handles.append(tb["handles"][index])
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
        return [(bi, t, i) for bi, b in enumerate(self.batches) for t in self.type_order[bi] if predicate(t) for i in range(len(b[t]["refs"])) if b[t]["refs"][i][0] != -1]

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
