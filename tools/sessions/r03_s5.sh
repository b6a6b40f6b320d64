#!/bin/bash
set -u
O=gpurun_out/r03_s5; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
timeout 600 python tools/ab_scene.py ragdoll "default:" "touch1:BEPUHIP_CODE_TOUCH=1" "plain-rows:BEPUHIP_ROW_POLICY=0" "plain-rows+touch1:BEPUHIP_ROW_POLICY=0,BEPUHIP_CODE_TOUCH=1" "nt-rows:BEPUHIP_ROW_POLICY=1" "nt-rows+touch1:BEPUHIP_ROW_POLICY=1,BEPUHIP_CODE_TOUCH=1" 2>&1 | tee $O/ab_ragdoll.txt
timeout 600 python tools/ab_scene.py pile "cooperative:" "exclusive:BEPUHIP_COOPERATIVE=0" "exclusive+touch1:BEPUHIP_COOPERATIVE=0,BEPUHIP_CODE_TOUCH=1" 2>&1 | tee $O/ab_pile.txt
