#!/bin/bash
# round 4, session 13: shared-body records move as lane pairs (one 32-byte request per record instead of two 16-byte ones): parity, same-box A/B against the
# library of the commit before (tools/experiments/variants/libbepuhip_lonerecords.so), counters
set -u
O=gpurun_out/r04_s13; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
OLD=$GRAFT_REPO_ROOT/tools/experiments/variants/libbepuhip_lonerecords.so
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_schedule_fuzz.py tests/test_gpu_parity.py tests/test_gpu_lattice.py -m gpu -x -q > $O/pytest_split.txt 2>&1
echo "pytest rc=$?"; grep -v "$F" $O/pytest_split.txt | tail -4 | cut -c1-400
for scene in pile crowd; do
  for lib in paired lone paired lone; do
    if [ $lib = lone ]; then export BEPUHIP_LIB=$OLD; else unset BEPUHIP_LIB; fi
    BEPUHIP_ROW_POLICY=0 timeout 300 python tools/ab_scene.py $scene "$lib records:" 2>&1 | grep "ms/step" | tee -a $O/ab_records.txt
  done
done
unset BEPUHIP_LIB
for scene in pile crowd; do
  for lib in paired lone; do
    if [ $lib = lone ]; then export BEPUHIP_LIB=$OLD; else unset BEPUHIP_LIB; fi
    echo "== $scene, $lib records" | tee -a $O/pmc_traffic.txt
    timeout 600 python tools/pmc_traffic.py --kernel cluster_kernel -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child $scene 2>&1 | tail -3 | tee -a $O/pmc_traffic.txt
  done
done
unset BEPUHIP_LIB
for scene in pile crowd; do
  echo "== $scene" | tee -a $O/pmc_sq.txt
  timeout 600 python tools/pmc_sq.py -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child $scene 2>&1 | tail -40 | tee -a $O/pmc_sq.txt
done
echo "== headline" | tee -a $O/pmc_sq.txt
timeout 600 python tools/pmc_sq.py -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-prewarm --traffic-child main 2>&1 | tail -40 | tee -a $O/pmc_sq.txt
