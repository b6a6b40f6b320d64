# round 6, session 44: evidence on the -fno-slp-vectorize library: bench line, rocprofv3 kernel stats of the same command, widened scenes, fuzzers, soak
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s44
mkdir -p $O
(rocm-smi --showclocks 2>&1 | grep -v '^$' | head -12) > $O/rocm_smi.txt
timeout 900 python bench.py --full-report $O/bench_full.json > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-traffic > $O/bench_profiled.json 2> $O/rocprof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; head -5 $O/kernel_stats.csv; rm -rf $O/prof
cd $GRAFT_REPO_ROOT
timeout 600 python tools/perf_widened.py 2>&1 | grep -v "^$" | tee $O/widened.txt
timeout 400 python tools/fuzz_device.py 4401 240 2>&1 | tail -2 | tee $O/fuzz_device.txt
FUZZ_SPECIALISE=1 timeout 500 python tools/fuzz_device.py 4402 300 2>&1 | tail -2 | tee $O/fuzz_device_specialised.txt
timeout 400 python tools/fuzz_structural.py 4403 240 2>&1 | tail -2 | tee $O/fuzz_structural.txt
timeout 400 python tools/soak.py 3 240 2>&1 | tail -3 | tee $O/soak.txt
