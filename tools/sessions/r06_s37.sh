# round 6, session 37: slot table in LDS (whole-island plans) x integration in halves, same-box A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s38
mkdir -p $O
timeout 500 python tools/ab_scene.py ragdoll "global table, one piece:BEPUHIP_SLOT_TABLE_IN_LDS=0,BEPUHIP_SPLIT_INTEGRATION=0" "LDS table, one piece:BEPUHIP_SLOT_TABLE_IN_LDS=1,BEPUHIP_SPLIT_INTEGRATION=0" "LDS table, halves:BEPUHIP_SLOT_TABLE_IN_LDS=1,BEPUHIP_SPLIT_INTEGRATION=1" "global table, halves:BEPUHIP_SLOT_TABLE_IN_LDS=0,BEPUHIP_SPLIT_INTEGRATION=1" "global table, one piece again:BEPUHIP_SLOT_TABLE_IN_LDS=0,BEPUHIP_SPLIT_INTEGRATION=0" "LDS table, one piece again:BEPUHIP_SLOT_TABLE_IN_LDS=1,BEPUHIP_SPLIT_INTEGRATION=0" 2>&1 | grep -v "^$" | tee -a $O/ab_slot_table.txt
for scene in pile crowd; do
  timeout 500 python tools/ab_scene.py $scene "one piece:BEPUHIP_SPLIT_INTEGRATION=0" "halves:BEPUHIP_SPLIT_INTEGRATION=1" "one piece again:BEPUHIP_SPLIT_INTEGRATION=0" "halves again:BEPUHIP_SPLIT_INTEGRATION=1" 2>&1 | grep -v "^$" | tee -a $O/ab_slot_table.txt
done
BEPUHIP_SPLIT_INTEGRATION=0 PHASES=16 PASS=2 timeout 300 python tools/cluster_trace.py 2>&1 | head -14 | tee $O/headline_trace_phases_one_piece.txt
BEPUHIP_SPLIT_INTEGRATION=1 PHASES=16 PASS=2 timeout 300 python tools/cluster_trace.py 2>&1 | head -14 | tee $O/headline_trace_phases_halves.txt
