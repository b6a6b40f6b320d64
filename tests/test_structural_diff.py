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
