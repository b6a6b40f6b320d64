#!/bin/bash
# round 4, session 1: replay the open seed-81 fuzz sequence with on-the-spot triage
set -u
O=gpurun_out/r04_s1; mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
timeout 420 python tools/replay_fuzz_device.py 81 "" 330 $O/replay81.log 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -60 | tee $O/replay81.txt
