#!/bin/bash
# round 4, session 29: looking for a slow-class box (instruction fetch) to measure the final library there: the class is decided by a short run of the bench scene with
# plain rows; on a fast-class box the session ends at once
set -u
O=gpurun_out/r04_s29; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
line=$(BEPUHIP_ROW_POLICY=0 STEPS=150 WARM=100 timeout 200 python tools/perf_cluster.py clusters 2>&1 | grep ms/step)
echo "$line"
ms=$(echo "$line" | sed 's/.* \([0-9.]*\) ms\/step.*/\1/')
if python -c "import sys; sys.exit(0 if float('$ms') > 0.2 else 1)"; then echo "SLOW class"; else echo "FAST class: nothing to do"; exit 0; fi
for scene in ragdoll crowd pile; do
  STEPS=300 timeout 400 python tools/ab_scene.py $scene "plain rows:BEPUHIP_ROW_POLICY=0" "code touch:BEPUHIP_ROW_POLICY=2" "code touch, 768 threads:BEPUHIP_ROW_POLICY=2,BEPUHIP_SPLIT_THREADS=768" "nt rows:BEPUHIP_ROW_POLICY=1" 2>&1 | grep "ms/step" | tee -a $O/slowbox_policies.txt
done
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "launch policy" $O/bench.err | head -4
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_s29/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "valu_busy", r.get("valu_busy"), d["config"]["row_policy"][:30])
for k, v in d["connected_scenes"].items():
    rr = v["roofline"]
    print(k, v["ms_per_step"], rr["frac"], rr.get("valu_busy"), rr.get("traffic_over_compulsory_stream"))
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -2 | tee $O/pytest_tail.txt
