# GPU-box refresh (developer helper): gpu tests, bench line, rocprofv3 kernel stats of the same command, schedule timings. Everything lands in gpurun_out/final.
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
(rocm-smi --showmemorypartition --showcomputepartition --showclocks --showmemvendor 2>&1 | grep -v '^$') > $O/rocm_smi.txt  # which class of box this is
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests.log 2>&1; grep -E "passed|failed|error" $O/gpu_tests.log | tail -3
timeout 600 python bench.py --steps 100 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-traffic > $O/bench_under_rocprof.json 2> $O/rocprof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; head -6 $O/kernel_stats.csv
# configs[1], the 100k-box pile: split-island plan vs the launch-per-batch schedule, kernel stats, HBM traffic, per-item timeline
cd $GRAFT_REPO_ROOT
(echo "== split-island plan (default)"; timeout 300 python tools/perf_pile.py | tail -2; echo "== BEPUHIP_NO_SPLIT=1 (launch-per-batch, hipGraph)"; BEPUHIP_NO_SPLIT=1 timeout 300 python tools/perf_pile.py | tail -2;
 for t in 512 768 1024; do echo "== BEPUHIP_SPLIT_THREADS=$t"; BEPUHIP_SPLIT_THREADS=$t timeout 300 python tools/perf_pile.py | tail -1; done) > $O/pile_timing.txt 2>&1; cat $O/pile_timing.txt
cd /tmp
STEPS=50 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pile -o r -- python $GRAFT_REPO_ROOT/tools/perf_pile.py > /dev/null 2> $O/rocprof_pile.err
find $O/prof_pile -name "*kernel_stats.csv" -exec cp {} $O/pile_kernel_stats.csv \; ; head -4 $O/pile_kernel_stats.csv
STEPS=5 timeout 900 python $GRAFT_REPO_ROOT/tools/pmc_traffic.py --kernel cluster_kernel -- python $GRAFT_REPO_ROOT/tools/perf_pile.py > $O/pile_pmc_traffic.txt 2>&1; cat $O/pile_pmc_traffic.txt
cd $GRAFT_REPO_ROOT
SCENE=pile PASS=4 timeout 300 python tools/cluster_trace.py > $O/pile_item_timeline.txt 2>&1; head -16 $O/pile_item_timeline.txt
# the island schedule's register budgets on the bench scene, and what recolouring buys (8f-4)
(for t in 1024 768 512; do echo "== BEPUHIP_CLUSTER_THREADS=$t"; BEPUHIP_CLUSTER_THREADS=$t timeout 300 python tools/perf_recolour.py | sed -n 2p; done; timeout 300 python tools/perf_recolour.py) > $O/cluster_variants.txt 2>&1; cat $O/cluster_variants.txt
