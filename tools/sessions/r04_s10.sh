#!/bin/bash
# round 4, session 10: after the per-batch overflow wait came back (pile regression of session 9), substep events, out-of-line requirk: suite, A/B scenes, conserving modes,
# then the bench line as the driver runs it and the rocprofv3 kernel stats of the connected scenes
set -u
O=gpurun_out/r04_s10; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -6 | cut -c1-400
for scene in ragdoll pile crowd; do BEPUHIP_ROW_POLICY=0 timeout 200 python tools/ab_scene.py $scene "product:" 2>&1 | grep "ms/step" | tee -a $O/ab_scenes.txt; done
timeout 300 python tools/perf_conserving.py 2>&1 | tail -5 | tee $O/conserving.txt
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "launch policy" $O/bench.err | tail -6
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s10/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["cpu_baseline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "launch_us", r["avg_launch_us"], "policy", d["config"]["row_policy"][:40])
    print("cpu", c["value"], c["cores"], c.get("ideal_socket_bound", {}).get("value"), [(e["threads"], round(e["value"]/1e6,1), round(e["parallel_efficiency"],2)) for e in c["thread_curve"]])
    for k, v in d["connected_scenes"].items():
        print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"], v["roofline"].get("traffic_over_compulsory_stream"), v["roofline"].get("traffic_detail", {}).get("write_bytes"))
    for e in d["scale_sweep"]["sizes"]:
        print("sweep", e["ragdolls"], e["constraints"], round(e["ms_per_step"],4), round(e["value"]/1e9,2), "G", e["clusters"], round(e["clusters_per_cu"],2), round(e["working_set_bytes"]/1e6), "MB", e.get("frac"), e.get("traffic_over_compulsory_stream"))
    w = d.get("widened_types"); print("widened", w and (w.get("ms_per_step"), w.get("schedule")))
    print({k: round(v, 3) for k, v in d["boundary"].items() if k.endswith("_ms")})
except Exception as e:
    print("bench parse failed", e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-traffic --no-connected-scenes ) > $O/prof_bench.json 2> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_pile -o pile -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child pile ) > /dev/null 2>> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_crowd -o crowd -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child crowd ) > /dev/null 2>> $O/prof.err
for f in $(find $O/prof $O/prof_pile $O/prof_crowd -name "*kernel_stats.csv"); do echo $f; head -2 $f | tail -1 | awk -F'",' '{print $2}' | cut -c1-100; done
