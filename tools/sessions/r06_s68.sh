# round 6, session 68: more fuzz seeds on the shipped library (the round's last GPU minutes)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s68
mkdir -p $O
timeout 500 python tools/fuzz_device.py 6801 360 2>&1 | tail -2 | tee $O/fuzz_device.txt
timeout 500 python tools/fuzz_structural.py 6802 360 2>&1 | tail -2 | tee $O/fuzz_structural.txt
timeout 300 python tools/fuzz_bounds.py 6803 120 2>&1 | tail -2 | tee $O/fuzz_bounds.txt
