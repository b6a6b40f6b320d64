"""Seeded small-scene generators for the parity tests (pure numpy + the Python SceneBuilder mirror)."""
from __future__ import annotations

import math

import numpy as np

from bepuphysics2_amd.scene import TYPE_TABLE, Scene, SceneBuilder, make_body
from bepuphysics2_amd.synthetic import joint_prestep, rand_quat, spring, unit  # noqa: F401  (the generators below and the tests use them under these names)

TWO_PI = 6.283185307179586
FLOAT_MAX = float(np.finfo(np.float32).max)


def random_dynamic_body(rng, position, speed=0.5):
    inv_mass = 1.0 / rng.uniform(1.0, 5.0)
    d = rng.uniform(0.5, 3.0, size=3) * inv_mass
    # diagonal local inverse inertia plus a small symmetric off-diagonal part (still positive definite)
    off = rng.uniform(-0.05, 0.05, size=3) * inv_mass
    inertia = (d[0], off[0], d[1], off[1], off[2], d[2])
    return make_body(position=position, orientation=rand_quat(rng), linear=rng.uniform(-speed, speed, 3), angular=rng.uniform(-speed, speed, 3),
                     inverse_inertia=inertia, inverse_mass=inv_mass)


def kinematic_body(rng, position, angular=(0, 0, 0.25)):
    return make_body(position=position, orientation=rand_quat(rng), linear=(0, 0, 0), angular=angular, inverse_inertia=(0,) * 6, inverse_mass=0.0)


def contact_prestep(rng, n, two_body, pos_a, pos_b=None, friction=1.0, freq=30.0, max_recovery=2.0):
    """ContactN[OneBody]PrestepData lane (ContactConvexTypes.cs:1418-1430)."""
    lane = []
    normal = unit(rng)
    for _ in range(n):
        lane += list(rng.uniform(-0.5, 0.5, 3).astype(np.float32)) + [np.float32(rng.uniform(-0.01, 0.02))]
    if two_body:
        lane += list((np.asarray(pos_b, np.float32) - np.asarray(pos_a, np.float32)))
    lane += list(normal)
    lane += [np.float32(friction)] + spring(freq, 1.0) + [np.float32(max_recovery)]
    return lane


def nonconvex_contact_prestep(rng, n, two_body, pos_a, pos_b=None, friction=1.0, freq=30.0, max_recovery=2.0):
    """ContactNNonconvex[OneBody]PrestepData lane (ContactNonconvexTypes.cs:58-66, :161-167): material, [OffsetB], n x {Offset, Depth, Normal}."""
    lane = [np.float32(friction)] + spring(freq, 1.0) + [np.float32(max_recovery)]
    if two_body:
        lane += list((np.asarray(pos_b, np.float32) - np.asarray(pos_a, np.float32)))
    for _ in range(n):
        lane += list(rng.uniform(-0.5, 0.5, 3).astype(np.float32)) + [np.float32(rng.uniform(-0.01, 0.02))] + list(unit(rng))
    return lane


def prestep_for(rng, type_id, pos_a, pos_b):
    nb, _, _, name = TYPE_TABLE[type_id]
    if name.startswith("Contact"):
        n = int(name[7])
        if "Nonconvex" in name:
            return nonconvex_contact_prestep(rng, n, nb == 2, pos_a, pos_b)
        return contact_prestep(rng, n, nb == 2, pos_a, pos_b)
    return joint_prestep(rng, type_id)


def random_graph_scene(seed, body_count, constraint_count, type_ids, kinematic_fraction=0.05, unconstrained_extra=3, warm=True) -> Scene:
    """Random constraint graph over random bodies: exercises batching, kinematic references, every given type."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder()
    positions = rng.uniform(-3, 3, size=(body_count + unconstrained_extra, 3)).astype(np.float32)
    for i in range(body_count + unconstrained_extra):
        if i < body_count and rng.random() < kinematic_fraction:
            sb.add_body(kinematic_body(rng, positions[i]))
        else:
            sb.add_body(random_dynamic_body(rng, positions[i]))
    added = 0
    while added < constraint_count:
        t = int(type_ids[rng.integers(len(type_ids))])
        nb = TYPE_TABLE[t][0]
        hs = list(rng.choice(body_count, size=nb, replace=False))
        if all(sb.is_kinematic(h) for h in hs):
            continue
        pa = positions[hs[0]]
        pb = positions[hs[1]] if nb >= 2 else None
        sb.add_constraint(t, hs, prestep_for(rng, t, pa, pb))
        added += 1
    scene = sb.build()
    if warm:  # nonzero warm-start impulses, as a running simulation would have
        for b in scene.batches:
            for tb in b:
                lanes = rng.uniform(0.0, 0.05, size=(tb.count, tb.impulse_floats)).astype(np.float32)
                from bepuphysics2_amd.scene import to_aosoa
                tb.accumulated[...] = to_aosoa(lanes, scene.bundle_width)
    return scene


def box_stack_scene(levels=3, per_level=4) -> Scene:
    """Small analytic box pyramid: Contact4 two-body between rows, Contact4OneBody against the ground (PyramidDemo geometry, Demos/Demos/PyramidDemo.cs:26-47)."""
    sb = SceneBuilder()
    mat = [np.float32(1.0)] + spring(30.0, 1.0) + [np.float32(2.0)]
    rows = []
    for r in range(levels):
        n = per_level - r
        row = []
        for c in range(n):
            x = -n / 2.0 + c + 0.5 * 0
            inv = (6.0, 0, 6.0, 0, 0, 6.0)  # unit cube mass 1: inverse inertia 1/(m/6)
            row.append(sb.add_body(make_body(position=(x + 0.5 * r, r + 0.5, 0), inverse_inertia=inv, inverse_mass=1.0)))
        rows.append(row)
    for h in rows[0]:  # ground contacts
        lane = []
        for dx, dz in ((-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5)):
            lane += [dx, -0.5, dz, 0.0]
        lane += [0, 1, 0] + mat
        sb.add_constraint(3, [h], lane)
    for r in range(1, levels):
        for c, h in enumerate(rows[r]):
            for below in (rows[r - 1][c], rows[r - 1][c + 1]):
                pa = sb._bodies[h][4:7]
                pb = sb._bodies[below][4:7]
                x0, x1 = max(pa[0], pb[0]) - 0.5, min(pa[0], pb[0]) + 0.5
                lane = []
                for x, z in ((x0, -0.5), (x1, -0.5), (x0, 0.5), (x1, 0.5)):
                    lane += [x - pa[0], -0.5, z, 0.0]
                lane += list(pb - pa) + [0, 1, 0] + mat
                sb.add_constraint(7, [h, below], lane)
    return sb.build()


def island_scene(seed, islands, bodies_per_island, constraints_per_island, type_ids, kinematic_shared=True) -> Scene:
    """Many small independent islands (each a random connected-ish graph) plus one kinematic body shared by all of them:
    the shape the island-per-workgroup (cluster) schedule is built for."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder()
    kin = sb.add_body(kinematic_body(rng, (0, 8, 0))) if kinematic_shared else None
    for isl in range(islands):
        base = rng.uniform(-20, 20, 3)
        hs, pos = [], []
        for _ in range(bodies_per_island):
            p = (base + rng.uniform(-1, 1, 3)).astype(np.float32)
            pos.append(p)
            hs.append(sb.add_body(random_dynamic_body(rng, p)))
        for k in range(constraints_per_island):
            t = int(type_ids[rng.integers(len(type_ids))])
            nb = TYPE_TABLE[t][0]
            if nb == 1:
                a = int(rng.integers(bodies_per_island))
                sb.add_constraint(t, [hs[a]], prestep_for(rng, t, pos[a], None))
            elif nb > 2:  # three- / four-body constraints: distinct bodies of the island, now and then with the shared kinematic body among them
                picks = [hs[int(x)] for x in rng.choice(bodies_per_island, size=nb, replace=False)]
                if kinematic_shared and rng.random() < 0.1:
                    picks[int(rng.integers(nb))] = kin
                sb.add_constraint(t, picks, prestep_for(rng, t, None, None))
            else:
                a, b = [int(x) for x in rng.choice(bodies_per_island, size=2, replace=False)]
                if kinematic_shared and rng.random() < 0.1:
                    pair, pb = ([hs[a], kin], np.asarray([0, 8, 0], np.float32)) if rng.random() < 0.5 else ([kin, hs[a]], pos[a])
                    pa = pos[a] if pair[0] == hs[a] else np.asarray([0, 8, 0], np.float32)
                    sb.add_constraint(t, pair, prestep_for(rng, t, pa, pb))
                else:
                    sb.add_constraint(t, [hs[a], hs[b]], prestep_for(rng, t, pos[a], pos[b]))
    for _ in range(5):  # a few unconstrained bodies
        sb.add_body(random_dynamic_body(rng, rng.uniform(-5, 5, 3)))
    scene = sb.build()
    from bepuphysics2_amd.scene import to_aosoa
    for b in scene.batches:
        for tb in b:
            lanes = rng.uniform(0.0, 0.05, size=(tb.count, tb.impulse_floats)).astype(np.float32)
            tb.accumulated[...] = to_aosoa(lanes, scene.bundle_width)
    return scene


def star_scene(seed, spokes=80, hubs=2, type_ids=(7, 22, 4, 47, 30), fallback_batch_threshold=64, extra_constraints=40) -> Scene:
    """Hub bodies with more constraints than the batch limit: every constraint on a hub needs its own batch, so the ones beyond
    FallbackBatchThreshold land in the sequential fallback batch (Solver.cs:1878-1884, TypeProcessor.cs:451-560). Plus a few hub-free constraints
    between spokes and one-body contacts so that the fallback batch is not the only thing that happens."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder(fallback_batch_threshold=fallback_batch_threshold)
    hub_handles = [sb.add_body(random_dynamic_body(rng, rng.uniform(-1, 1, 3))) for _ in range(hubs)]
    spoke_handles = [sb.add_body(random_dynamic_body(rng, rng.uniform(-4, 4, 3)) if i % 9 else kinematic_body(rng, rng.uniform(-4, 4, 3))) for i in range(spokes)]
    for i, s in enumerate(spoke_handles):
        t = type_ids[i % len(type_ids)]
        h = hub_handles[i % hubs]
        pair = [h, s] if i % 3 else [s, h]
        sb.add_constraint(t, pair, prestep_for(rng, t, sb._bodies[pair[0]][4:7], sb._bodies[pair[1]][4:7]))
    for _ in range(extra_constraints):
        a, b = rng.choice(len(spoke_handles), 2, replace=False)
        ha, hb = spoke_handles[a], spoke_handles[b]
        if sb.is_kinematic(ha) and sb.is_kinematic(hb):
            continue
        t = type_ids[int(rng.integers(len(type_ids)))]
        sb.add_constraint(t, [ha, hb], prestep_for(rng, t, sb._bodies[ha][4:7], sb._bodies[hb][4:7]))
    for s in spoke_handles[::7]:
        if not sb.is_kinematic(s):
            sb.add_constraint(3, [s], prestep_for(rng, 3, sb._bodies[s][4:7], None))
    return sb.build()


def concat_scenes(a: Scene, b: Scene) -> Scene:
    """Both scenes side by side in one simulation: b's bodies follow a's, batch k holds both scenes' batch k (a body never appears twice in a batch, because the
    two scenes share no bodies), same-type type batches joined lane after lane."""
    from bepuphysics2_amd.scene import TypeBatchData, to_aosoa
    w = a.bundle_width
    assert b.bundle_width == w
    na, ha = a.body_count, int(a.handle_to_index.size)
    bodies = np.concatenate([a.bodies, b.bodies])
    i2h = np.concatenate([a.index_to_handle, b.index_to_handle + ha]).astype(np.int32)
    h2i = np.concatenate([a.handle_to_index, np.where(b.handle_to_index >= 0, b.handle_to_index + na, b.handle_to_index)]).astype(np.int32)
    batches = []
    for k in range(max(len(a.batches), len(b.batches))):
        merged = {}
        for scene, offset in ((a, 0), (b, na)):
            if k >= len(scene.batches):
                continue
            for tb in scene.batches[k]:
                refs = tb.refs_lanes(w).copy()
                refs = np.where(refs >= 0, refs + offset, refs)  # the kinematic flag (bit 30) rides along: indices are far below it
                merged.setdefault(tb.type_id, []).append((refs, tb.prestep_lanes(w), tb.accumulated_lanes(w)))
        out = []
        for type_id, parts in merged.items():
            refs, pre, acc = (np.concatenate([p[i] for p in parts]) for i in range(3))
            out.append(TypeBatchData(type_id, refs.shape[0], to_aosoa(refs.astype(np.int32), w, fill=-1), to_aosoa(pre.astype(np.float32), w), to_aosoa(acc.astype(np.float32), w)))
        batches.append(out)
    kin = np.concatenate([a.constrained_kinematic_handles, b.constrained_kinematic_handles + ha]).astype(np.int32)
    return Scene(bodies, i2h, h2i, batches, kin, w)
