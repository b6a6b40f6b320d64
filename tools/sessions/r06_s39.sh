# round 6, session 39: GPU suite + fuzzers on the library with the slot table in LDS and the integration in halves on split plans
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s39
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 400 python tools/fuzz_device.py 3901 240 2>&1 | tail -4 | tee $O/fuzz_device.txt
timeout 400 python tools/fuzz_structural.py 3902 240 2>&1 | tail -4 | tee $O/fuzz_structural.txt
BEPUHIP_SPLIT_INTEGRATION=1 timeout 400 python tools/fuzz_device.py 3903 180 2>&1 | tail -4 | tee $O/fuzz_device_halves_everywhere.txt
