#!/bin/bash
# round 4, session 3: the overflow-wait fix. The regression test on the product library and on the library with round 3's wait put back (must fail there), rate statistics
# with and without schedule fuzzing, cold processes, then the whole GPU suite.
set -u
O=gpurun_out/r04_s3; mkdir -p $O
export TMPDIR=/tmp
OLD=tools/experiments/overflow_race/libbepuhip_oldwait.so
F='^HIP\|^ROCm\|^Hostname\|^Librccl'
echo "== regression test, product library"; timeout 300 python -m pytest tests/test_gpu_schedule_fuzz.py -k overflow -q 2>&1 | grep -v "$F" | tail -4 | tee $O/regression_product.txt
echo "== regression test, round 3's wait (expected to FAIL)"; BEPUHIP_LIB=$OLD timeout 300 python -m pytest tests/test_gpu_schedule_fuzz.py -k overflow -q 2>&1 | grep -v "$F" | tail -6 | cut -c1-400 | tee $O/regression_oldwait.txt
echo "== rates, product library"; timeout 300 python tools/race_hunt.py 81 91 150 default jitter jitter_mode0 jitter_mode1 2>&1 | grep -v "$F" | grep "runs differ" | tee $O/rates_product.txt
echo "== rates, round 3's wait"; BEPUHIP_LIB=$OLD timeout 300 python tools/race_hunt.py 81 91 150 default jitter jitter_mode0 jitter_mode1 2>&1 | grep -v "$F" | grep "runs differ" | tee $O/rates_oldwait.txt
echo "== cold processes, product library"; for i in 1 2 3 4 5 6 7 8; do timeout 100 python tools/race_hunt.py 81 91 3 default graph ivk its1 sub1 mode1 2>&1 | grep "runs differ" | awk '{s+=$2; n+=$4} END {print "cold process: " s " of " n " runs differ"}'; done | tee $O/cold_product.txt
echo "== cold processes, round 3's wait"; for i in 1 2 3 4; do BEPUHIP_LIB=$OLD timeout 100 python tools/race_hunt.py 81 91 3 default graph ivk its1 sub1 mode1 2>&1 | grep "runs differ" | awk '{s+=$2; n+=$4} END {print "cold process: " s " of " n " runs differ"}'; done | tee $O/cold_oldwait.txt
echo "== GPU suite"; timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -8
