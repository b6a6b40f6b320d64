#!/bin/bash
# round 4, session 30: on a slow-class box only: how many 8 KB spans of code touch do the split plans want?
set -u
O=gpurun_out/r04_s30; mkdir -p $O
export TMPDIR=/tmp
line=$(BEPUHIP_ROW_POLICY=0 STEPS=150 WARM=100 timeout 200 python tools/perf_cluster.py clusters 2>&1 | grep ms/step)
echo "$line"
ms=$(echo "$line" | sed 's/.* \([0-9.]*\) ms\/step.*/\1/')
if python -c "import sys; sys.exit(0 if float('$ms') > 0.2 else 1)"; then echo "SLOW class"; else echo "FAST class: nothing to do"; exit 0; fi
for scene in crowd pile ragdoll; do
  STEPS=300 timeout 400 python tools/ab_scene.py $scene "one span:BEPUHIP_ROW_POLICY=2,BEPUHIP_CODE_TOUCH=1" "two spans:BEPUHIP_ROW_POLICY=2,BEPUHIP_CODE_TOUCH=2" "three spans:BEPUHIP_ROW_POLICY=2,BEPUHIP_CODE_TOUCH=3" "one span again:BEPUHIP_ROW_POLICY=2,BEPUHIP_CODE_TOUCH=1" 2>&1 | grep "ms/step" | tee -a $O/code_touch_spans_slowbox.txt
done
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-scale-sweep > $O/bench_quick.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_s30/bench_quick.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "valu_busy", r.get("valu_busy"), r.get("traffic_detail", {}).get("issue", {}).get("waves_per_simd"))
for k, v in d["connected_scenes"].items():
    rr = v["roofline"]; print(k, v["ms_per_step"], rr.get("valu_busy"))
PY
