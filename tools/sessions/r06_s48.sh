# round 6, session 48: end_constraints of the headline scene phase by phase (BEPUHIP_PLAN_STATS=2)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s48
mkdir -p $O
BEPUHIP_PLAN_STATS=2 timeout 300 python tools/perf_upload.py 15000 5 > $O/perf_upload_stats.txt 2>&1; tail -60 $O/perf_upload_stats.txt | cut -c1-400
