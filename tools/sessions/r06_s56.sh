# round 6, session 56: the profiling pass times the launch with events the launch carries itself (hipExtLaunchKernel): GPU suite, bench line, rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s56
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --full-report $O/bench_full.json > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/s56/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["boundary"]["end_constraints_ms"], d["connected_scenes"]["pile_100k"]["ms_per_step"], d["connected_scenes"]["ragdoll_crowd"]["ms_per_step"], d["widened_types"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-traffic > $O/bench_profiled.json 2> $O/rocprof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; head -4 $O/kernel_stats.csv | cut -c1-200; rm -rf $O/prof
