#!/bin/bash
set -u
O=gpurun_out/r04_s8b; mkdir -p $O
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | grep -v "^\.\|^$" | tail -30 | cut -c1-300
grep -v "$F" $O/pytest_gpu.txt | tail -3
timeout 120 python tools/fuzz_device.py 5001 60 2>&1 | grep -v "$F" | tail -2 | cut -c1-400
