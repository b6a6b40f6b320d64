#!/bin/bash
# round 4, session 31: the library as committed at the end of the round: full GPU suite, smoke, the bench line as the driver runs it
set -u
O=gpurun_out/r04_s31; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -4 | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -1 | cut -c1-200
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_s31/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "valu_busy", r.get("valu_busy"), d["config"]["row_policy"][:30])
for k, v in d["connected_scenes"].items():
    rr = v["roofline"]
    print(k, v["ms_per_step"], rr["frac"], rr.get("valu_busy"), rr.get("traffic_over_compulsory_stream"), v.get("upload_ms"))
print({k: round(v, 3) for k, v in d["boundary"].items() if k.endswith("_ms")})
PY
grep "cluster plan (host)" $O/bench.err | head -4
