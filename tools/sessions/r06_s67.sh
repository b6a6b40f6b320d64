# round 6, session 67: the shipped library again after the not-shipped launch experiment: soak (three threads, 240 s), GPU suite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s67
mkdir -p $O
timeout 400 python tools/soak.py 3 240 2>&1 | grep -i "complaint\|soak:" | tail -4 | tee $O/soak.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
