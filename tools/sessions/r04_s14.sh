#!/bin/bash
# round 4, session 14: the crowd test fails with paired record accesses: which half (loads / stores), is it flaky, does the library of the commit before pass here
set -u
O=gpurun_out/r04_s14; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
V=$GRAFT_REPO_ROOT/tools/experiments/variants
tools/probes/pair_record_probe.bin 2>&1 | tee $O/pair_probe.txt
T="tests/test_gpu_split.py::test_ragdoll_crowd_is_split_and_bit_exact"
run() { echo "== $1" | tee -a $O/crowd_test.txt; shift; env "$@" timeout 300 python -m pytest $T -m gpu -x -q 2>&1 | grep -v "$F" | grep "passed\|failed\|bodies_max_ulp\|Error" | cut -c1-300 | tee -a $O/crowd_test.txt; }
run "lone records (commit before)" BEPUHIP_LIB=$V/libbepuhip_lonerecords.so
run "paired" X=1
run "paired again" X=1
run "paired loads, lone stores" BEPUHIP_LIB=$V/libbepuhip_lonestores.so
run "lone loads, paired stores" BEPUHIP_LIB=$V/libbepuhip_loneloads.so
run "paired, no local hand-offs" BEPUHIP_SPLIT_LOCAL_HANDOFF=0
run "paired, 12 clusters" BEPUHIP_SPLIT_CLUSTERS=12
run "paired, 2 clusters" BEPUHIP_SPLIT_CLUSTERS=2
