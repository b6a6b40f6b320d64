#!/bin/bash
# Builds libbepuhip_oldwait.so: today's library with round 3's per-batch publish counters put back into wait_predecessors' overflow paths (oldwait.patch) — the
# race tools/fuzz_device.py seed 81 found — so that the regression tests can be shown to FAIL on it (tests/test_gpu_schedule_fuzz.py passes on the product library).
# CPU only (hipcc cross-compiles); the result ships to the GPU box with the snapshot. Usage: tools/experiments/overflow_race/make_oldwait_lib.sh
set -e
HERE=$(cd $(dirname $0) && pwd); REPO=$(cd $HERE/../../.. && pwd)
T=${TMPDIR:-/tmp}/bepu_oldwait; rm -rf $T; mkdir -p $T/bepuphysics2_amd $T/include
cp -r $REPO/bepuphysics2_amd/csrc $T/bepuphysics2_amd/csrc; rm -rf $T/bepuphysics2_amd/csrc/build $T/bepuphysics2_amd/csrc/*.so
cp $REPO/include/bepuhip.h $T/include/; cp $REPO/bepuphysics2_amd/build.py $T/bepuphysics2_amd/; touch $T/bepuphysics2_amd/__init__.py
( cd $T/bepuphysics2_amd/csrc && patch -p0 < $HERE/oldwait.patch )
( cd $T && python -c "from bepuphysics2_amd import build; build.build_hip()" )
cp $T/bepuphysics2_amd/csrc/libbepuhip.so $HERE/libbepuhip_oldwait.so
echo "built $HERE/libbepuhip_oldwait.so"
