#!/bin/bash
# round 4, session 17: asm stores carry their own wait states (the crowd failure of session 13 was a store-data hazard the compiler cannot see inside asm);
# paired records: probe, suite, same-box A/B against the commit before, counters, the bench line as the driver runs it, rocprofv3 kernel stats
set -u
O=gpurun_out/r04_s17; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
OLD=$GRAFT_REPO_ROOT/tools/experiments/variants/libbepuhip_lonerecords.so
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
timeout 300 tools/probes/pair_pingpong_probe.bin 2>&1 | tee $O/pair_pingpong.txt
timeout 300 python -m pytest "tests/test_gpu_split.py::test_ragdoll_crowd_is_split_and_bit_exact" -m gpu -x -q 2>&1 | grep -v "$F" | tail -2
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "crowd test still fails: stopping"; exit 1; fi
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -6 | cut -c1-400
for scene in pile crowd; do
  for lib in paired lone paired lone; do
    if [ $lib = lone ]; then export BEPUHIP_LIB=$OLD; else unset BEPUHIP_LIB; fi
    BEPUHIP_ROW_POLICY=0 timeout 300 python tools/ab_scene.py $scene "$lib records:" 2>&1 | grep "ms/step" | tee -a $O/ab_records.txt
  done
done
unset BEPUHIP_LIB
BEPUHIP_PLAN_STATS=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "launch policy" $O/bench.err | tail -4
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s17/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["cpu_baseline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "launch_us", r["avg_launch_us"], "policy", d["config"]["row_policy"][:40])
    print("cpu", c["value"], c["cores"], c.get("ideal_socket_bound", {}).get("value"))
    for k, v in d["connected_scenes"].items():
        print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"], v["roofline"].get("traffic_over_compulsory_stream"), v["roofline"].get("traffic_detail", {}).get("write_bytes"))
    for e in d["scale_sweep"]["sizes"]:
        print("sweep", e["ragdolls"], e["constraints"], round(e["ms_per_step"],4), round(e["value"]/1e9,2), "G", e["clusters"], e.get("frac"))
    w = d.get("widened_types"); print("widened", w and (w.get("ms_per_step"), w.get("schedule")))
    print({k: round(v, 3) for k, v in d["boundary"].items() if k.endswith("_ms")})
    print({k: (round(v.get("ms_per_step", 0), 4) if isinstance(v, dict) else v) for k, v in d.get("lattice", {}).items()})
except Exception as e:
    print("bench parse failed", e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-traffic --no-connected-scenes ) > $O/prof_bench.json 2> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_pile -o pile -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child pile ) > /dev/null 2>> $O/prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_crowd -o crowd -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child crowd ) > /dev/null 2>> $O/prof.err
for f in $(find $O/prof $O/prof_pile $O/prof_crowd -name "*kernel_stats.csv"); do echo $f; head -2 $f | tail -1 | cut -c1-160; done
