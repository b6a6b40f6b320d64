# round 6, session 50: is the 10.9 us between two back-to-back solves the end-of-kernel write-back of dirty L2 lines? The same trace with non-temporal rows (BEPUHIP_ROW_POLICY=1)
# and with the plain rows again (0), generic hot unit both (BENCH_SPECIALISE=0: the exact-type unit has no non-temporal twin)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s50
mkdir -p $O
for policy in 1 0; do
  BENCH_SPECIALISE=0 BEPUHIP_ROW_POLICY=$policy timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof$policy -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-traffic --no-connected-scenes > $O/bench_policy$policy.json 2> $O/rocprof$policy.err
  T=$(find $O/prof$policy -name "*kernel_trace.csv" | head -1)
  echo "row policy $policy"; python $GRAFT_REPO_ROOT/tools/trace_durations.py $T cluster_kernel 100 | sed -n 1,8p
  rm -rf $O/prof$policy
done | tee $O/gaps_by_policy.txt
