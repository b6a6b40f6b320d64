#!/bin/bash
# round 4, session 4: fuzzers on the fixed library (fresh seeds, every second scene under schedule fuzzing), then the bench line with the scale sweep
set -u
O=gpurun_out/r04_s4; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
for seed in 1001 1002 1003; do timeout 200 python tools/fuzz_device.py $seed 115 $O/fuzz_device_$seed.log 2>&1 | grep -v "$F" | tail -3 | cut -c1-600 | tee -a $O/fuzz_device.txt; done
for seed in 2001 2002; do timeout 200 python tools/fuzz_structural.py $seed 80 2>&1 | grep -v "$F" | tail -2 | cut -c1-600 | tee -a $O/fuzz_structural.txt; done
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "launch policy" $O/bench.err | tail -4
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s4/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["cpu_baseline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "policy", d["config"]["row_policy"][:40], "ws", r.get("working_set_bytes"))
    print("cpu", c["value"], c["cores"], c.get("placement", {}).get("first_socket_physical_cores"), c.get("ideal_socket_bound", {}).get("value"), [(e["threads"], round(e["value"]/1e6,1), round(e["parallel_efficiency"],2), round(e["solve_worker_busy_fraction"],2), round(e["solve_work_inflation_vs_one_thread"],2)) for e in c["thread_curve"]])
    for k, v in d["connected_scenes"].items():
        print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"], v["roofline"].get("traffic_over_compulsory_stream"))
    sw = d.get("scale_sweep")
    if sw and "sizes" in sw:
        for e in sw["sizes"]:
            print("sweep", e["ragdolls"], e["constraints"], round(e["ms_per_step"],4), round(e["value"]/1e9,2), "G", e["clusters"], round(e["clusters_per_cu"],2), round(e["working_set_bytes"]/1e6), "MB", e.get("frac"), e.get("traffic_over_compulsory_stream"), round(e["memory_stream_frac_of_peak"],3), round(e["upload_ms"]))
    else:
        print("sweep", sw)
    w = d.get("widened_types")
    print("widened", w and (w.get("ms_per_step"), w.get("value"), w.get("schedule")))
    b = d["boundary"]
    print({k: v for k, v in b.items() if k.endswith("_ms")})
except Exception as e:
    print("bench parse failed", e)
PY
