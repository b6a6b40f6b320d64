#!/bin/bash
set -u
O=gpurun_out/r03_s4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_parity.py tests/test_gpu_structural.py -x -q -m gpu > $O/pytest_split.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_split.txt
timeout 600 python tools/ab_scene.py pile "prefetch-x4:" "no-prefetch:BEPUHIP_PREFETCH=0" "prefetch-x4,plain-launch:BEPUHIP_COOPERATIVE=0" "no-prefetch,plain-launch:BEPUHIP_PREFETCH=0,BEPUHIP_COOPERATIVE=0" 2>&1 | tee $O/ab_pile.txt
BEPUHIP_PLAN_STATS=1 STEPS=5 timeout 300 python tools/ab_scene.py pile "x:" 2>&1 | grep -i "split plan" | tee -a $O/ab_pile.txt
