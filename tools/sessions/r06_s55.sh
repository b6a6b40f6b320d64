# round 6, session 55: solves without the event pair (bepuhip_set_solve_timing off by default), no tail workgroups when the plan owns every body: GPU suite, bench line,
# the gap between two solves in a kernel trace, pile / crowd on the cooperative path
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s55
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --full-report $O/bench_full.json > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/s55/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["boundary"]["end_constraints_ms"], d["connected_scenes"]["pile_100k"]["ms_per_step"], d["connected_scenes"]["ragdoll_crowd"]["ms_per_step"], d["widened_types"])
PY
for scene in pile crowd; do echo -n "$scene (cooperative launch): "; STEPS=300 timeout 300 python tools/ab_scene.py $scene "x:" 2>&1 | grep "ms/step" | cut -c1-110; done | tee $O/ab_split_cooperative.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-traffic --no-connected-scenes > $O/bench_traced.json 2> $O/rocprof.err
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_durations.py $T cluster_kernel 100 | tee $O/gaps.txt | sed -n 1,8p
rm -rf $O/prof
