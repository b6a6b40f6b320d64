# round 6, session 42: split plans at 8 / 12 / 16 waves on units compiled for the scene's exact types, without the SLP vectoriser's register pairs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s42
mkdir -p $O
export AB_SPECIALISE=1 BEPUHIP_SPLIT_ALLOW_1024=1
for scene in pile crowd; do
  timeout 900 python tools/ab_scene.py $scene "default:" "512:BEPUHIP_SPLIT_THREADS=512" "768:BEPUHIP_SPLIT_THREADS=768" "1024:BEPUHIP_SPLIT_THREADS=1024" "default again:" "1024 again:BEPUHIP_SPLIT_THREADS=1024" 2>&1 | grep -v "^$" | tee -a $O/ab_split_threads.txt
done
