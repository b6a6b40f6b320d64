#!/bin/bash
# round 4, session 33: the incremental contact update starts before the barrier that ends the previous substep's last sweep (row loads in flight across the barrier):
# parity on the scenes with contacts, then same-box A/B against the library of the commit before
set -u
O=gpurun_out/r04_s33; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
OLD=$GRAFT_REPO_ROOT/tools/experiments/variants/libbepuhip_before.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -3
for scene in ragdoll crowd pile; do
  for lib in new before new before; do
    if [ $lib = before ]; then export BEPUHIP_LIB=$OLD; else unset BEPUHIP_LIB; fi
    STEPS=300 timeout 300 python tools/ab_scene.py $scene "$lib:" 2>&1 | grep "ms/step" | tee -a $O/ab_incremental_before_barrier.txt
  done
done
