"""CPU stand-in of one rank of a split scene (TEST INFRASTRUCTURE): the oracle solves the share, the hook reproduces the device's snapshot / delta / apply arithmetic."""
from __future__ import annotations

import numpy as np


class OracleShare:
    """CPU stand-in (tests only): the oracle solves the share; the hook reproduces the device's snapshot / delta / apply arithmetic in float32.
    The oracle integrates a body inside its first constraint's warm start, so the substep's snapshot is the velocity call-back applied to the
    last synchronised velocity (the same float32 operations as Demos/DemoCallbacks.cs:100-109)."""

    def __init__(self, share: Share, dt: float, solve_description, callbacks, exchange: BoundaryExchange):
        self.share, self.dt, self.sd, self.cb, self.exchange = share, dt, solve_description, callbacks, exchange
        sub_dt = np.float32(np.float32(dt) / np.float32(solve_description.substep_count))
        lin = np.float32(min(max(1.0 - callbacks.linear_damping, 0.0), 1.0))
        ang = np.float32(min(max(1.0 - callbacks.angular_damping, 0.0), 1.0))
        self.lin_damp = np.float32(np.power(lin, sub_dt, dtype=np.float32))
        self.ang_damp = np.float32(np.power(ang, sub_dt, dtype=np.float32))
        self.gravity_dt = (np.asarray(callbacks.gravity, dtype=np.float32) * sub_dt).astype(np.float32)
        self.snapshot = None

    def _velocities(self):
        b = self.share.scene.bodies[self.share.boundary_local]
        return np.concatenate([b[:, 8:11], b[:, 12:15]], axis=1).astype(np.float32)

    def _integrated(self, v):
        out = v.copy()
        out[:, 0:3] = (v[:, 0:3] + self.gravity_dt) * self.lin_damp
        out[:, 3:6] = v[:, 3:6] * self.ang_damp
        return out

    def hook(self, _substep, pass_index):
        if pass_index == 0:
            self.snapshot = self._integrated(self.synced)
        v = self._velocities()
        new = self.snapshot + self.exchange.reduce(v - self.snapshot)
        bodies = self.share.scene.bodies
        bodies[self.share.boundary_local, 8:11] = new[:, 0:3]
        bodies[self.share.boundary_local, 12:15] = new[:, 3:6]
        self.snapshot = new
        self.synced = new

    def solve(self, oracle_solve, frames: int = 1, threads: int = 1):
        for _ in range(frames):
            self.synced = self._velocities()
            oracle_solve(self.share.scene, self.dt, self.sd, self.cb, threads=threads, exchange=self.hook)


def solve_oracle_shares_in_process(shares, dt, solve_description, callbacks, frames: int = 1, threads: int = 1):
    """The CPU lattice: every share through OracleShare, one thread per rank, the per-pass averaged exchange through the same ThreadExchange the in-process GPU harness
    uses (same rank order of the sum, same division by the holders). What tests/test_gpu_lattice.py holds the device's block-Jacobi shares against."""
    import threading

    import oracle_ffi
    from bepuphysics2_amd.lattice import ThreadExchange
    ex = ThreadExchange(shares)
    errors = []

    class RankExchange:
        def __init__(self, rank):
            self.rank = rank

        def reduce(self, local):
            return ex.reduce(self.rank, np.ascontiguousarray(local, dtype=np.float32), False)

    def run(rank):
        try:
            OracleShare(shares[rank], dt, solve_description, callbacks, RankExchange(rank)).solve(oracle_ffi.solve, frames=frames, threads=threads)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            ex.barrier.abort()

    workers = [threading.Thread(target=run, args=(r,)) for r in range(len(shares))]
    for t in workers:
        t.start()
    for t in workers:
        t.join()
    if errors:
        raise errors[0]
    return ex
