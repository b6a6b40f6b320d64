#!/bin/bash
set -u
O=gpurun_out/r03_s9; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_parity.py tests/test_gpu_colouring.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/ab_scene.py pile "local hand-off:BEPUHIP_COOPERATIVE=0" "records only:BEPUHIP_COOPERATIVE=0,BEPUHIP_SPLIT_LOCAL_HANDOFF=0" 2>&1 | tee $O/ab_pile.txt
timeout 600 python tools/ab_scene.py crowd "local hand-off:BEPUHIP_COOPERATIVE=0" "records only:BEPUHIP_COOPERATIVE=0,BEPUHIP_SPLIT_LOCAL_HANDOFF=0" 2>&1 | tee $O/ab_crowd.txt
timeout 300 python tools/fuzz_device.py 77 45 2>&1 | tail -2
RAGDOLLS=15000 timeout 300 python tools/perf_recolour.py 2>&1 | head -3 | tee $O/colouring.txt
