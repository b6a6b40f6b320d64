"""Per-item timeline of the first cluster of the island-per-workgroup schedule (GPU box). Kernel-tuning aid, not part of the product."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks

ragdolls = int(os.environ.get("RAGDOLLS", "15000"))
if os.environ.get("SCENE", "ragdolls") == "pile":  # BASELINE.json configs[1]: one island, the split plan
    sim = HostSimulation.scene("pile", int(os.environ.get("BOXES", "100000")), 0, 0, 5)
elif os.environ.get("SCENE") == "crowd":  # the ragdolls lying on each other: one island of 240,000 bodies, the split plan with joints and contacts
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 2, 5)
else:
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 0, 5)
scene, sd = sim.export(), sim.solve_description()
cb = PoseIntegratorCallbacks()
s = HipSolver()
s.upload(scene)
for _ in range(2):
    s.solve(1 / 60, sd, cb)
s.set_cluster_trace(True)
s.solve(1 / 60, sd, cb)
passes = int((1 + sd.iterations()).sum())
tr = s.cluster_trace(passes)
# 0x80 / 0x81: a merged manifold group (one-body / two-body Contact1..4 lanes of one batch in one wave); its members leave no record of their own
names = {0x80: "CMo", 0x81: "CM", 0: "C1o", 1: "C2o", 2: "C3o", 3: "C4o", 4: "C1", 5: "C2", 6: "C3", 7: "C4", 22: "Ball", 23: "AHinge", 25: "Swing", 26: "TServo", 27: "TLimit", 30: "AMotor", 31: "Weld", 46: "Swivel", 47: "Hinge"}
t_first = int(tr[..., 0][tr[..., 0] > 0].min())
print(f"items per pass: {tr.shape[1]} ({int((tr[0][:, 0] > 0).sum())} claimed as items of their own), passes: {passes}")
for p in range(passes):
    rec = tr[p]
    rec = rec[rec[:, 0] > 0]
    st, en = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    span = int(en.max() - st.min())
    busy = int((en - st).sum())
    print(f"pass {p}: begins at {int(st.min()) - t_first:8d} cyc, span {span:7d} cyc, sum of item times {busy:8d} ({busy / span:.2f} waves busy incl. waits), mean item {busy / len(st):.0f}")
print(f"frame span (first claim to last publish): {int(tr[..., 1].max()) - t_first} cyc")
if os.environ.get("PHASES"):  # between the sweeps: per wave the clock at the top of a substep, after its incremental items, behind that barrier, behind the integration's barrier
    full = s.cluster_trace(128)
    waves = int(os.environ.get("PHASES"))
    for sub in range(int(sd.substep_count)):
        ph = full[128 - 16 + sub].reshape(-1)[: waves * 4].reshape(waves, 4).astype(np.int64)
        top, mid, end, own = ph[:, 0], ph[:, 1], ph[:, 2], ph[:, 3]
        print(f"substep {sub}: top of substep {int(top.min()) - t_first:8d} .. {int(top.max()) - t_first:8d} | incremental items done (per wave, 0 = none) {[int(x - top.min()) if x else 0 for x in own]} | barrier behind them {int(mid.max() - top.min()):6d} | integration + barrier {int(end.max() - mid.max()):6d} cyc")
p = int(os.environ.get("PASS", "1"))
rec = tr[p]
order = [k for k in np.argsort(rec[:, 0]) if rec[k, 0] > 0]
t0 = rec[rec[:, 0] > 0][:, 0].min()
print(f"--- pass {p} timeline (cycles from pass start): item batch type wave start dur count | loads setup wait tail")
for k in order:
    meta = int(rec[k, 2])
    print(f"{k:4d} b{(meta >> 16) & 0xFFFF:<3d} {names.get((meta >> 8) & 0xFF, '?'):7s} w{meta & 0xFF} {int(rec[k, 0] - t0):7d} {int(rec[k, 1] - rec[k, 0]):6d} {int(rec[k, 3]):3d} | {int(rec[k, 4] - rec[k, 0]):5d} {int(rec[k, 5] - rec[k, 4]):5d} {int(rec[k, 6] - rec[k, 5]):5d} {int(rec[k, 1] - rec[k, 6]):5d}")
