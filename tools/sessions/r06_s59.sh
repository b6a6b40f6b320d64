# round 6, session 59: fuzzers and soak on the final library (no event pair around a solve, no tail workgroups when the plan owns every body)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s59
mkdir -p $O
timeout 400 python tools/fuzz_device.py 5901 240 2>&1 | tail -2 | tee $O/fuzz_device.txt
FUZZ_SPECIALISE=1 timeout 400 python tools/fuzz_device.py 5902 200 2>&1 | tail -2 | tee $O/fuzz_device_specialised.txt
timeout 400 python tools/fuzz_structural.py 5903 240 2>&1 | tail -2 | tee $O/fuzz_structural.txt
timeout 400 python tools/soak.py 3 200 2>&1 | tail -3 | tee $O/soak.txt
