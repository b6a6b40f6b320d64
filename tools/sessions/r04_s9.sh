#!/bin/bash
# round 4, session 9: the round's measurement session — GPU suite, the bench line as the driver runs it (PMC child runs, cpu baseline, scale sweep), rocprofv3 kernel
# stats of the same commands, fuzzers, the side tools whose numbers DESIGN.md quotes
set -u
O=gpurun_out/r04_s9; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -6 | cut -c1-400
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "launch policy" $O/bench.err | tail -6
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s9/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["cpu_baseline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "launch_us", r["avg_launch_us"], "policy", d["config"]["row_policy"][:40], "ws", r.get("working_set_bytes"))
    print("cpu", c["value"], c["cores"], c.get("ideal_socket_bound", {}).get("value"), [(e["threads"], round(e["value"]/1e6,1), round(e["parallel_efficiency"],2)) for e in c["thread_curve"]])
    for k, v in d["connected_scenes"].items():
        print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"], v["roofline"].get("traffic_over_compulsory_stream"), v["roofline"].get("traffic_detail", {}).get("write_bytes"))
    for e in d["scale_sweep"]["sizes"]:
        print("sweep", e["ragdolls"], e["constraints"], round(e["ms_per_step"],4), round(e["value"]/1e9,2), "G", e["clusters"], round(e["clusters_per_cu"],2), round(e["working_set_bytes"]/1e6), "MB", e.get("frac"), e.get("traffic_over_compulsory_stream"))
    w = d.get("widened_types"); print("widened", w and (w.get("ms_per_step"), w.get("schedule")))
    print({k: round(v, 3) for k, v in d["boundary"].items() if k.endswith("_ms")})
    print("lattice", {k: (v.get("velocity_err_max"), v.get("within_north_star_tolerance")) for k, v in d["lattice"].items() if isinstance(v, dict)})
except Exception as e:
    print("bench parse failed", e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-traffic --no-connected-scenes ) > $O/prof_bench.json 2> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_pile -o pile -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child pile ) > /dev/null 2>> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_crowd -o crowd -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child crowd ) > /dev/null 2>> $O/prof.err
find $O/prof $O/prof_pile $O/prof_crowd -name "*kernel_stats.csv" | head; for f in $(find $O/prof $O/prof_pile $O/prof_crowd -name "*kernel_stats.csv"); do echo $f; head -4 $f | cut -c1-260; done
timeout 200 python tools/fuzz_device.py 4001 75 2>&1 | grep -v "$F" | tail -2 | cut -c1-400 | tee $O/fuzz_device.txt
timeout 200 python tools/fuzz_structural.py 4002 60 2>&1 | grep -v "$F" | tail -2 | cut -c1-400 | tee $O/fuzz_structural.txt
timeout 300 python tools/perf_widened.py 15000 4000 2>&1 | tail -4 | tee $O/widened.txt
timeout 300 python tools/perf_conserving.py 2>&1 | tail -5 | tee $O/conserving.txt
(BEPUHIP_PLAN_STATS=2 timeout 400 python tools/perf_churn.py pile 40 2>&1 | grep "flush\|pile" | tail -4) | tee $O/churn.txt
(for f in "" "--lattice-exact"; do timeout 300 python bench.py --lattice $f --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e9,2), 'G', d['config']['sharding'][-95:])"; done) | tee $O/lattice.txt
