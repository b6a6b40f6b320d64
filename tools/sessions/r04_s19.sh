#!/bin/bash
# round 4, session 19: fresh fuzz seeds on the library with paired records (every second scene under schedule fuzzing), the bench line with its new SQ counter pass, smoke
set -u
O=gpurun_out/r04_s19; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
timeout 300 python tools/fuzz_device.py 7001 120 $O/fuzz_device_7001.log 2>&1 | grep -v "$F" | tail -2 | cut -c1-400 | tee $O/fuzz_device.txt
timeout 300 python tools/fuzz_device.py 7002 120 $O/fuzz_device_7002.log 2>&1 | grep -v "$F" | tail -2 | cut -c1-400 | tee -a $O/fuzz_device.txt
timeout 300 python tools/fuzz_structural.py 7003 120 2>&1 | grep -v "$F" | tail -2 | cut -c1-400 | tee $O/fuzz_structural.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -3 | tee $O/smoke.txt
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s19/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "valu_busy", r.get("valu_busy"), "issue", r.get("traffic_detail", {}).get("issue"))
    for k, v in d["connected_scenes"].items():
        rr = v["roofline"]
        print(k, v["ms_per_step"], rr["frac"], rr.get("valu_busy"), rr.get("traffic_over_compulsory_stream"), rr.get("traffic_detail", {}).get("write_bytes"), rr.get("traffic_detail", {}).get("issue", {}).get("waves_per_simd"))
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $O/bench.err | cut -c1-300
