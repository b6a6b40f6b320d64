# GPU-box refresh (developer helper): gpu tests, bench line, rocprofv3 kernel stats of the same command, schedule timings.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(rocm-smi --showmemorypartition --showcomputepartition --showclocks --showmemvendor 2>&1 | grep -v '^$') > gpurun_out/final/rocm_smi.txt  # which class of box this is
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 100 --warmup 20 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -3 gpurun_out/final/bench.err; cat gpurun_out/final/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-traffic > $GRAFT_REPO_ROOT/gpurun_out/final/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/final/rocprof.err
cat $GRAFT_REPO_ROOT/gpurun_out/final/bench_under_rocprof.json
find $GRAFT_REPO_ROOT/gpurun_out/final/prof -name "*kernel_stats.csv" -exec head -5 {} \;
cd $GRAFT_REPO_ROOT
STEPS=300 WARM=200 timeout 300 python tools/perf_cluster.py waves 2>&1 | tail -8
timeout 300 python tools/perf_pile.py 2>&1 | tail -3
