#!/bin/bash
set -u
O=gpurun_out/r04_s8; mkdir -p $O
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -8 | cut -c1-400
