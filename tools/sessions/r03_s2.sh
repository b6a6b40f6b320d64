#!/bin/bash
# Round 3, GPU session 2: the split-plan changes (row prefetch through LDS-DMA, cooperative launch, event-number epochs) and the code-touch switch: parity first, then A/B.
set -u
O=gpurun_out/r03_s2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_parity.py tests/test_gpu_structural.py tests/test_gpu_edges.py -x -q -m gpu > $O/pytest_split.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_split.txt
timeout 600 python tools/ab_scene.py pile "prefetch+coop:" "no-prefetch:BEPUHIP_PREFETCH=0" "prefetch,plain-launch:BEPUHIP_COOPERATIVE=0" "neither:BEPUHIP_PREFETCH=0,BEPUHIP_COOPERATIVE=0" "prefetch+touch1:BEPUHIP_CODE_TOUCH=1" > $O/ab_pile.txt 2>&1
cat $O/ab_pile.txt
timeout 600 python tools/ab_scene.py crowd "default:" "no-prefetch:BEPUHIP_PREFETCH=0" "plain-launch:BEPUHIP_COOPERATIVE=0" > $O/ab_crowd.txt 2>&1
cat $O/ab_crowd.txt
timeout 600 python tools/ab_scene.py ragdoll "default:" "touch1:BEPUHIP_CODE_TOUCH=1" "touch2:BEPUHIP_CODE_TOUCH=2" "plain-rows:BEPUHIP_ROW_POLICY=0" "plain-rows+touch1:BEPUHIP_ROW_POLICY=0,BEPUHIP_CODE_TOUCH=1" "nt-rows+touch1:BEPUHIP_ROW_POLICY=1,BEPUHIP_CODE_TOUCH=1" > $O/ab_ragdoll.txt 2>&1
cat $O/ab_ragdoll.txt
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1; head -3 $O/rocm_smi.txt
