#!/bin/bash
set -u
O=gpurun_out/r03_s6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -4
BEPUHIP_PLAN_STATS=1 timeout 600 python tools/ab_scene.py ragdoll "policy:" "plain:BEPUHIP_ROW_POLICY=0" "touch1:BEPUHIP_ROW_POLICY=2" 2>&1 | grep -v "split plan\|^bepuhip end_constraints\|^  *[a-z].* ms$" | tee $O/ab_ragdoll.txt
BEPUHIP_PLAN_STATS=1 timeout 600 python tools/ab_scene.py pile "policy:" "plain:BEPUHIP_ROW_POLICY=0" "touch1:BEPUHIP_ROW_POLICY=2" "touch2:BEPUHIP_ROW_POLICY=3" 2>&1 | grep "ms/step\|launch policy" | tee $O/ab_pile.txt
BEPUHIP_PLAN_STATS=1 timeout 600 python tools/ab_scene.py crowd "policy:" "plain:BEPUHIP_ROW_POLICY=0" "touch1:BEPUHIP_ROW_POLICY=2" 2>&1 | grep "ms/step\|launch policy" | tee $O/ab_crowd.txt
