"""The diff that reconstructs a frame's structural changes from a type batch's handles and references (host/bepu_host.cpp DiffTypeBatch, the C++ twin of
integration/csharp/HipTimestepper.cs), on its own: random append / swap-with-last / reference-patch histories, the emitted operations replayed on a plain list."""
import numpy as np

from bepuphysics2_amd import hostlib

W = 8


def to_aosoa(lanes: np.ndarray, fields: int) -> np.ndarray:
    n = lanes.shape[0]
    out = np.zeros((max((n + W - 1) // W, 1), fields, W), dtype=lanes.dtype)
    for i in range(n):
        out[i // W, :, i % W] = lanes[i]
    return out.reshape(-1)


def replay(ops, payload, device, bodies, prestep_floats):
    for kind, batch, type_id, index, slot, reference, offset, _ in ops:
        if kind == 0:
            assert index == len(device)  # AllocateInTypeBatch appends
            device.append((payload[offset:offset + bodies].view(np.int32).copy(), payload[offset + bodies:offset + bodies + prestep_floats].view(np.float32).copy()))
        elif kind == 1:  # TypeProcessor.Remove: the last constraint takes the index
            device[index] = device[-1]
            device.pop()
        elif kind == 2:
            device[index][0][slot] = reference
        elif kind == 3:
            device[index], device[slot] = device[slot], device[index]
        else:
            raise AssertionError(kind)


def test_diff_reproduces_any_history_of_appends_removals_and_reference_patches():
    rng = np.random.default_rng(11)
    kinds = np.zeros(4, dtype=np.int64)
    for trial in range(600):
        bodies, pf = int(rng.integers(1, 5)), int(rng.integers(1, 6))
        handles = [int(h) for h in rng.choice(1000, size=int(rng.integers(0, 60)), replace=False)]
        refs = {h: rng.integers(0, 500, size=bodies).astype(np.int32) for h in handles}
        pre = {h: rng.random(pf).astype(np.float32) for h in handles}
        old_refs = np.array([refs[h] for h in handles], dtype=np.int32).reshape(-1, bodies)
        current = list(handles)
        for _ in range(int(rng.integers(0, 40))):  # the reference's own moves, in an order the diff never sees
            if current and rng.random() < 0.5:
                i = int(rng.integers(len(current)))
                current[i] = current[-1]
                current.pop()
            else:
                h = 1000 + len(refs)
                refs[h], pre[h] = rng.integers(0, 500, size=bodies).astype(np.int32), rng.random(pf).astype(np.float32)
                current.append(h)
        new_refs = {h: refs[h].copy() for h in current}
        for h in current:
            if rng.random() < 0.1:  # a body moved in memory (UpdateForBodyMemoryMove)
                new_refs[h][int(rng.integers(bodies))] = int(rng.integers(500, 600))
        ops, payload = hostlib.diff_type_batch(3, 7, bodies, pf, handles, old_refs, current,
                                               to_aosoa(np.array([new_refs[h] for h in current], dtype=np.int32).reshape(-1, bodies), bodies),
                                               to_aosoa(np.array([pre[h] for h in current], dtype=np.float32).reshape(-1, pf), pf))
        assert all(op[1] == 3 and op[2] == 7 for op in ops)
        device = [(refs[h].copy(), pre[h].copy()) for h in handles]
        replay(ops, payload, device, bodies, pf)
        assert len(device) == len(current)
        for i, h in enumerate(current):
            assert np.array_equal(device[i][0], new_refs[h]) and np.array_equal(device[i][1], pre[h]), (trial, i)
        for op in ops:
            kinds[op[0]] += 1
    assert (kinds > 100).all(), kinds  # additions, removals, reference patches and swaps all occurred


def test_an_unchanged_type_batch_costs_no_operations():
    handles = np.arange(40, dtype=np.int32)
    refs = np.arange(80, dtype=np.int32).reshape(40, 2)
    ops, _ = hostlib.diff_type_batch(0, 4, 2, 3, handles, refs, handles, to_aosoa(refs, 2), to_aosoa(np.zeros((40, 3), np.float32), 3))
    assert ops.shape[0] == 0


def test_a_reused_constraint_handle_is_a_removal_and_an_addition():
    """Solver.HandlePool hands freed handles out again last-in-first-out (IdPool.Take): Remove(h) + Add(...) in one frame returns h for ANOTHER constraint, possibly at the
    same index of the same type batch. The diff tells the two apart by the handles of their bodies; what keeps handle AND bodies is a survivor (its device
    state travels with it), a body that only moved in memory is a reference patch. Random histories with a LIFO pool; operations replayed on a plain list; removals first."""
    rng = np.random.default_rng(23)
    reused_total = same_index_total = 0
    for trial in range(400):
        bodies, pf = int(rng.integers(1, 3)), int(rng.integers(1, 5))
        body_index = {b: b for b in range(400)}  # body handle -> memory index (moves when a body is removed from the middle)
        pool, next_handle = [], 0
        live = []  # (constraint handle, body handles, prestep)

        def add():
            nonlocal next_handle
            if pool:
                h = pool.pop()
            else:
                h, next_handle = next_handle, next_handle + 1
            live.append((h, tuple(int(x) for x in rng.choice(400, size=bodies, replace=False)), rng.random(pf).astype(np.float32)))

        def remove(i):
            pool.append(live[i][0])
            live[i] = live[-1]
            live.pop()

        for _ in range(int(rng.integers(1, 50))):
            add()
        old = list(live)
        old_index = dict(body_index)
        for _ in range(int(rng.integers(1, 30))):
            r = rng.random()
            if live and r < 0.45:
                remove(int(rng.integers(len(live))))
            elif live and r < 0.6:
                remove(len(live) - 1)  # remove the last ...
                add()                  # ... and add: same handle, same index
            else:
                add()
        for b in rng.choice(400, size=5, replace=False):  # a few bodies moved in memory
            body_index[int(b)] = 1000 + int(b)
        refs = lambda entries, index: np.array([[index[b] for b in e[1]] for e in entries], dtype=np.int32).reshape(-1, bodies)
        handles = lambda entries: np.array([e[0] for e in entries], dtype=np.int32)
        body_handles = lambda entries: np.array([e[1] for e in entries], dtype=np.int32).reshape(-1, bodies)
        ops, payload, survivor = hostlib.diff_type_batch_identities(
            2, 9, bodies, pf, handles(old), refs(old, old_index), body_handles(old), handles(live), to_aosoa(refs(live, body_index), bodies), body_handles(live),
            to_aosoa(np.array([e[2] for e in live], dtype=np.float32).reshape(-1, pf), pf))
        kinds = [int(op[0]) for op in ops]
        assert kinds == sorted(kinds, key=lambda k: k != 1), "every removal comes before anything else"
        device = [(refs([e], old_index)[0].copy(), e[2].copy(), ("old", j)) for j, e in enumerate(old)]
        tagged = []
        for j, (r, p, tag) in enumerate(device):
            tagged.append([r, p, tag])
        for kind, _, _, index, slot, reference, offset, _ in ops:  # replay, keeping track of which old constraint's device state sits where
            if kind == 0:
                assert index == len(tagged)
                tagged.append([payload[offset:offset + bodies].view(np.int32).copy(), payload[offset + bodies:offset + bodies + pf].view(np.float32).copy(), ("new", None)])
            elif kind == 1:
                tagged[index] = tagged[-1]
                tagged.pop()
            elif kind == 2:
                tagged[index][0][slot] = reference
            else:
                tagged[index], tagged[slot] = tagged[slot], tagged[index]
        assert len(tagged) == len(live)
        old_by_identity = {(e[0], e[1]): j for j, e in enumerate(old)}
        for i, e in enumerate(live):
            assert np.array_equal(tagged[i][0], refs([e], body_index)[0]), (trial, i)
            was = old_by_identity.get((e[0], e[1]), -1)
            assert survivor[i] == was, (trial, i, survivor[i], was)
            if was >= 0:
                assert tagged[i][2] == ("old", was), "a survivor keeps the device state it had"
            else:
                assert tagged[i][2] == ("new", None) and np.array_equal(tagged[i][1], e[2])
                reused = e[0] in {o[0] for o in old}
                reused_total += reused
                same_index_total += reused and i < len(old) and old[i][0] == e[0]
    assert reused_total > 200 and same_index_total > 50, (reused_total, same_index_total)
