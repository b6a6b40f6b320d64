#!/bin/bash
# round 4, session 32: fresh fuzz seeds on the library as committed (parallel split planner, two spans of code touch on split plans)
set -u
O=gpurun_out/r04_s32; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
timeout 200 python tools/fuzz_device.py 8001 70 2>&1 | grep -v "$F" | tail -1 | cut -c1-400 | tee $O/fuzz.txt
BEPUHIP_ROW_POLICY=2 timeout 200 python tools/fuzz_device.py 8002 70 2>&1 | grep -v "$F" | tail -1 | cut -c1-400 | tee -a $O/fuzz.txt
timeout 200 python tools/fuzz_structural.py 8003 70 2>&1 | grep -v "$F" | tail -1 | cut -c1-400 | tee -a $O/fuzz.txt
