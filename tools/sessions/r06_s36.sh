# round 6, session 36: the integration in two halves beside the incremental contact update (BEPUHIP_SPLIT_INTEGRATION), same-box A/B + parity + the phase stamps
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s36
mkdir -p $O
for scene in ragdoll pile crowd; do
  timeout 500 python tools/ab_scene.py $scene "one piece:BEPUHIP_SPLIT_INTEGRATION=0" "halves:BEPUHIP_SPLIT_INTEGRATION=1" "one piece again:BEPUHIP_SPLIT_INTEGRATION=0" "halves again:BEPUHIP_SPLIT_INTEGRATION=1" 2>&1 | grep -v "^$" | tee -a $O/ab_split_integration.txt
done
PHASES=16 PASS=2 timeout 300 python tools/cluster_trace.py 2>&1 | head -14 | tee $O/headline_trace_phases.txt
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -5 | tee $O/pytest_gpu.txt
