#!/bin/bash
set -u
O=gpurun_out/r03_s8; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest_gpu.txt | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL"
timeout 300 python tools/fuzz_device.py 31 70 2>&1 | tail -3 | tee $O/fuzz_device.txt
timeout 300 python tools/fuzz_structural.py 2>&1 | tail -3 | tee $O/fuzz_structural.txt
BEPUHIP_PLAN_STATS=2 timeout 600 python tools/boundary_probe.py 2>&1 | grep "union-find\|cluster plan\|_ms" | tail -3
