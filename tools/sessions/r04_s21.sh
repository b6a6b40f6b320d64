#!/bin/bash
# round 4, session 21: row touch (a wave without items left asks for the rows it will want after the coming barrier): same-process A/B on the three bench scenes
set -u
O=gpurun_out/r04_s21; mkdir -p $O
export TMPDIR=/tmp
for scene in ragdoll pile crowd; do
  BEPUHIP_ROW_POLICY=0 STEPS=300 timeout 400 python tools/ab_scene.py $scene "row touch:" "off:BEPUHIP_ROW_TOUCH=0" "row touch again:" "off again:BEPUHIP_ROW_TOUCH=0" 2>&1 | grep "ms/step\|bodies" | tee -a $O/ab_row_touch.txt
done
