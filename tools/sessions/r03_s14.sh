#!/bin/bash
set -u
O=gpurun_out/r03_s14; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest_gpu.txt | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl"
BEPUHIP_PLAN_STATS=1 timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "end_constraints\|launch policy" $O/bench.err | tail -24
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03_s14/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["cpu_baseline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "policy", d["config"]["row_policy"][:40])
    print("cpu", c["value"], c["cores"], [(e["threads"], round(e["value"]/1e6,1), round(e["parallel_efficiency"],2), round(e["solve_worker_busy_fraction"],2), round(e["solve_work_inflation_vs_one_thread"],2)) for e in c["thread_curve"]])
    for k, v in d["connected_scenes"].items():
        print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["algorithmic_frac_of_peak"], v["roofline"]["traffic"])
    w = d.get("widened_types")
    print("widened", w and (w["ms_per_step"], w["value"], w["schedule"]))
    b = d["boundary"]
    print({k: v for k, v in b.items() if k.endswith("_ms")})
except Exception as e:
    print("bench parse failed", e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-traffic --no-connected-scenes ) > $O/prof_bench.json 2> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_pile -o pile -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child pile ) > /dev/null 2>> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_crowd -o crowd -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child crowd ) > /dev/null 2>> $O/prof.err
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
timeout 300 python tools/ab_scene.py ragdoll "plain:BEPUHIP_ROW_POLICY=0" "touch1:BEPUHIP_ROW_POLICY=2" 2>&1 | tee $O/box_class.txt
(BEPUHIP_PLAN_STATS=2 timeout 500 python tools/perf_churn.py pile 60 2>&1 | grep "flush\|pile" | tail -4; BEPUHIP_PLAN_STATS=2 timeout 500 python tools/perf_churn.py crowd 60 2>&1 | grep "flush\|crowd" | tail -4) | tee $O/churn.txt
timeout 300 python tools/perf_widened.py 15000 4000 2>&1 | tail -4 | tee $O/widened.txt
timeout 300 python tools/perf_conserving.py 2>&1 | tail -5 | tee $O/conserving.txt
(for f in "" "--lattice-no-clusters" "--lattice-exact"; do timeout 300 python bench.py --lattice $f --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e9,2), 'G', d['config']['sharding'][-95:])"; done) | tee $O/lattice.txt
timeout 200 python tools/fuzz_structural.py 61 60 2>&1 | tail -1 | tee $O/fuzz_structural.txt
timeout 200 python tools/fuzz_device.py 62 60 2>&1 | tail -1 | tee $O/fuzz_device.txt
