# round 6, session 63: the launch-per-batch schedule on the headline scene under a kernel trace: kernels' own time against the step's period (what the launches cost)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s63
mkdir -p $O
cat > /tmp/lpb.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
scene, sd = bench.build_scene(15000, 5)
cb = PoseIntegratorCallbacks()
s = HipSolver(use_clusters=False)
s.upload(scene)
for _ in range(30): s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
t0 = time.perf_counter()
for _ in range(100): s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
print(f"launch-per-batch, hipGraph replay: {(time.perf_counter() - t0) * 10:.4f} ms/step")
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python /tmp/lpb.py 2>&1 | grep -v "^\[W\|^W2" | grep -v "^    @" | tail -12 | tee $O/launch_per_batch.txt
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - <<PY | tee -a $O/launch_per_batch.txt
import csv
rows = list(csv.DictReader(open("$S")))
tot = 0
for r in rows[:8]:
    print(r["Name"][:70].replace("(anonymous namespace)::", ""), r["Calls"], f'{float(r["AverageNs"]) / 1e3:.2f} us avg', f'{float(r["TotalDurationNs"]) / 130 / 1e3:.1f} us per step')
tr = sorted(csv.DictReader(open("$T")), key=lambda r: int(r["Start_Timestamp"]))
tr = tr[len(tr) // 2:]  # the timed half
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr)
span = int(tr[-1]["End_Timestamp"]) - int(tr[0]["Start_Timestamp"])
print(f"second half of the trace: {len(tr)} launches, kernels busy {busy / 1e6:.2f} ms of {span / 1e6:.2f} ms ({busy / span:.2f}); mean gap {(span - busy) / len(tr) / 1e3:.2f} us per launch")
PY
rm -rf $O/prof
