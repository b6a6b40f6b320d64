#!/bin/bash
# round 4, session 23: does the gap between consecutive step launches (wall 0.1629 ms against 0.1561 ms of kernel) shrink when the kernel's stores leave L2 while it runs?
set -u
O=gpurun_out/r04_s23; mkdir -p $O
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/tools/experiments/variants/libbepuhip_wtstores.so
for scene in ragdoll pile; do
  for lib in product wtstores product wtstores; do
    if [ $lib = wtstores ]; then export BEPUHIP_LIB=$V; else unset BEPUHIP_LIB; fi
    BEPUHIP_ROW_POLICY=0 STEPS=400 timeout 300 python tools/ab_scene.py $scene "$lib:" 2>&1 | grep "ms/step" | tee -a $O/ab_wt_stores.txt
  done
done
