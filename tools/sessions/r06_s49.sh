# round 6, session 49: the gap between two back-to-back solves (kernel trace of the bench's timed loop): end of launch n -> start of launch n + 1
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s49
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-traffic --no-connected-scenes > $O/bench_traced.json 2> $O/rocprof.err
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_durations.py $T cluster_kernel 50 | tee $O/gaps.txt
rm -rf $O/prof
python -c "
import json; d=json.load(open('$O/bench_traced.json')); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"
