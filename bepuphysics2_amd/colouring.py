"""Batch colouring on the device: the bulk counterpart of Solver.Add's first-fit batch walk (BepuPhysics/Solver.cs:984-1014) and of
BatchCompressor (BepuPhysics/BatchCompressor.cs:233). ``recolour_scene`` returns the same constraints regrouped into the batches the device computed; the host
(or the oracle, in the parity tests) must solve THAT scene — a different colouring applies a body's constraints in a different order."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from .native import _check, _ptr, load_library
from .scene import FALLBACK_BATCH_THRESHOLD, KINEMATIC_MASK, Scene, TypeBatchData, to_aosoa

ORDER_INSERTION, ORDER_LARGEST_DEGREE_FIRST = 0, 1


def colour_constraints(refs: np.ndarray, body_count: int, order: int = ORDER_INSERTION, fallback_batch_threshold: int = FALLBACK_BATCH_THRESHOLD,
                       device: int = 0) -> Tuple[np.ndarray, int, int]:
    """refs: int32 [count, <=4] encoded body references (-1 = unused). Returns (batch index per constraint, batch count, rounds the device needed)."""
    refs = np.asarray(refs, dtype=np.int32)
    padded = np.full((refs.shape[0], 4), -1, dtype=np.int32)
    padded[:, :refs.shape[1]] = refs
    colours = np.empty(refs.shape[0], dtype=np.int32)
    batches, rounds = C.c_int32(0), C.c_int32(0)
    lib = load_library()
    _check(lib, lib.bepuhip_colour_constraints(int(device), _ptr(padded), padded.shape[0], int(body_count), int(order), int(fallback_batch_threshold), _ptr(colours),
                                               C.byref(batches), C.byref(rounds)))
    return colours, int(batches.value), int(rounds.value)


def flatten_constraints(scene: Scene):
    """Every constraint of the scene in batch order: (type_id [n], refs [n, 4] padded with -1, prestep lanes, accumulated lanes) with per-constraint row lists."""
    w = scene.bundle_width
    types, refs, pre, acc = [], [], [], []
    for batch in scene.batches:
        for tb in batch:
            occupied = tb.occupied(w) if tb.count else np.zeros(0, dtype=bool)
            r = tb.refs_lanes(w)[occupied]
            padded = np.full((r.shape[0], 4), -1, dtype=np.int32)
            padded[:, :r.shape[1]] = r
            types.append(np.full(r.shape[0], tb.type_id, dtype=np.int32))
            refs.append(padded)
            pre.extend(tb.prestep_lanes(w)[occupied])
            acc.extend(tb.accumulated_lanes(w)[occupied])
    return np.concatenate(types) if types else np.zeros(0, np.int32), np.concatenate(refs) if refs else np.zeros((0, 4), np.int32), pre, acc


def max_dynamic_degree(scene: Scene) -> int:
    """The lower bound of any colouring: the largest number of constraints on one dynamic body."""
    _, refs, _, _ = flatten_constraints(scene)
    r = refs[(refs >= 0) & ((refs & KINEMATIC_MASK) == 0)]
    return int(np.bincount(r).max()) if r.size else 0


def recolour_scene(scene: Scene, order: int = ORDER_LARGEST_DEGREE_FIRST, device: int = 0) -> Tuple[Scene, int]:
    """The scene's constraints in the batches bepuhip_colour_constraints assigns; inside a (batch, type) the constraints keep their relative order."""
    from .scene import TYPE_TABLE
    w = scene.bundle_width
    types, refs, pre, acc = flatten_constraints(scene)
    colours, batch_count, rounds = colour_constraints(refs, scene.body_count, order, device=device)
    if batch_count > FALLBACK_BATCH_THRESHOLD:
        raise ValueError("the colouring needs the sequential fallback batch; recolour_scene only builds synchronized batches")
    batches: List[List[TypeBatchData]] = []
    for b in range(batch_count):
        row = []
        in_batch = np.nonzero(colours == b)[0]
        for type_id in sorted(set(types[in_batch].tolist())):
            idx = in_batch[types[in_batch] == type_id]
            nb = TYPE_TABLE[type_id][0]
            row.append(TypeBatchData(type_id, int(idx.size), to_aosoa(refs[idx][:, :nb].astype(np.int32), w, fill=-1),
                                     to_aosoa(np.stack([pre[i] for i in idx]).astype(np.float32), w), to_aosoa(np.stack([acc[i] for i in idx]).astype(np.float32), w)))
        batches.append(row)
    return Scene(scene.bodies.copy(), scene.index_to_handle.copy(), scene.handle_to_index.copy(), batches, scene.constrained_kinematic_handles.copy(), w), rounds
