#!/bin/bash
# round 4, session 15: paired records fail the crowd test only when loads AND stores are paired: timing, unit or protocol?
set -u
O=gpurun_out/r04_s15; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
T="tests/test_gpu_split.py::test_ragdoll_crowd_is_split_and_bit_exact"
run() { echo "== $1" | tee -a $O/crowd_test.txt; shift; env "$@" timeout 300 python -m pytest $T -m gpu -x -q 2>&1 | grep -v "$F" | grep "passed\|failed\|bodies_max_ulp.*Assert\|Error:" | cut -c1-220 | tee -a $O/crowd_test.txt; }
run "paired" X=1
run "paired, jitter 5" BEPUHIP_DEBUG_JITTER=5
run "paired, jitter 9" BEPUHIP_DEBUG_JITTER=9
run "paired, poll sleep 8" BEPUHIP_SHARED_POLL=8
run "paired, poll sleep 0" BEPUHIP_SHARED_POLL=0
run "paired, 768 threads" BEPUHIP_SPLIT_THREADS=768
run "paired, 1024 threads" BEPUHIP_SPLIT_THREADS=1024
run "paired, 256 threads" BEPUHIP_SPLIT_THREADS=256
run "paired, 64 threads" BEPUHIP_SPLIT_THREADS=64
run "paired, plain rows" BEPUHIP_ROW_POLICY=0
run "paired, nt rows" BEPUHIP_ROW_POLICY=1
run "paired, code touch" BEPUHIP_ROW_POLICY=2
run "paired, plain launch" BEPUHIP_COOPERATIVE=0
run "paired, 40 clusters" BEPUHIP_SPLIT_CLUSTERS=40
run "paired, 100 clusters" BEPUHIP_SPLIT_CLUSTERS=100
