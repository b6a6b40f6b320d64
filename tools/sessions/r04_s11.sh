#!/bin/bash
# round 4, session 11: several smaller clusters per CU (VERDICT r3 #4's co-residency question), the batched structural call's failure test, pile churn with the batched call
set -u
O=gpurun_out/r04_s11; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
rocm-smi --showclocks > $O/rocm_smi.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_structural.py -m gpu -x -q > $O/pytest_structural.txt 2>&1
echo "pytest rc=$?"; grep -v "$F" $O/pytest_structural.txt | tail -4 | cut -c1-400
BEPUHIP_ROW_POLICY=0 WARM=300 STEPS=400 timeout 900 python tools/perf_cluster.py coresident 2>&1 | grep "ms/step" | tee $O/coresident.txt
timeout 300 python tools/perf_churn.py pile 40 2>&1 | tail -12 | tee $O/churn_pile.txt
