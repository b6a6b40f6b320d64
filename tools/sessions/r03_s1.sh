#!/bin/bash
# Round 3, GPU session 1: LDS-DMA probe, the full GPU suite (with the new full-size crowd test), the bench line with the counter-based roofline and the
# persistent-session cpu_baseline, a kernel-trace profile of the same command.
set -u
mkdir -p gpurun_out/r03_s1
O=gpurun_out/r03_s1
export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $O/rocm_smi.txt 2>&1
nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt
( cd tools/probes && timeout 120 ./glds_probe.bin ) > $O/glds_probe.txt 2>&1
echo "glds rc=$?" >> $O/glds_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-traffic --no-connected-scenes ) > $O/prof_bench.json 2> $O/prof.err
echo "prof rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03_s1/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["cpu_baseline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "basis", r["basis"], "alg_frac", r["algorithmic_frac_of_peak"], "traffic", r["traffic"])
    print("cpu", c["value"], c["cores"], [(e["threads"], round(e["value"]/1e6,1), round(e["parallel_efficiency"],2)) for e in c["thread_curve"]])
    for k, v in d["connected_scenes"].items():
        print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["algorithmic_frac_of_peak"], v["roofline"]["traffic"])
    print("boundary", d["boundary"])
except Exception as e:
    print("bench parse failed", e)
PY
