# round 6, session 34: same-box A/B of the island kernels' row prefetch (BEPUHIP_ROW_PREFETCH = claims ahead) on the three bench scenes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s34
mkdir -p $O
(rocm-smi --showclocks 2>&1 | grep -v '^$' | head -20) > $O/rocm_smi.txt
for scene in ragdoll pile crowd; do
  timeout 500 python tools/ab_scene.py $scene "off:BEPUHIP_ROW_PREFETCH=0" "ahead16:BEPUHIP_ROW_PREFETCH=16" "ahead24:BEPUHIP_ROW_PREFETCH=24" "ahead32:BEPUHIP_ROW_PREFETCH=32" "ahead48:BEPUHIP_ROW_PREFETCH=48" "ahead12:BEPUHIP_ROW_PREFETCH=12" "off again:BEPUHIP_ROW_PREFETCH=0" 2>&1 | grep -v "^$" | tee -a $O/ab_row_prefetch_wave.txt
done

