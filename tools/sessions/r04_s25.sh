#!/bin/bash
# round 4, session 25: planner / poll switches on the split plans after the paired records (same process, bit-identical results required)
set -u
O=gpurun_out/r04_s25; mkdir -p $O
export TMPDIR=/tmp
for scene in crowd pile; do
  BEPUHIP_ROW_POLICY=0 STEPS=300 timeout 600 python tools/ab_scene.py $scene "default:" "separate items:BEPUHIP_SPLIT_SEPARATE=1" "poll sleep 0:BEPUHIP_SHARED_POLL=0" "poll sleep 3:BEPUHIP_SHARED_POLL=3" "224 clusters:BEPUHIP_SPLIT_CLUSTERS=224" "no refine:BEPUHIP_SPLIT_REFINE=0" "refine 4:BEPUHIP_SPLIT_REFINE=4" "default again:" 2>&1 | grep "ms/step\|bodies" | tee -a $O/split_switches.txt
done
