# round 6, session 73: HostTypeBatch::perm_inverse sized by the type batch: the failing scene alone, the GPU suite, two minutes of the structural fuzzer with the seed that found it
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s73
mkdir -p $O
timeout 100 python tools/probes/replay_fuzz_structural.py 6802 490 --load tests/golden/fuzz_structural_6802_490_state.json 2>&1 | grep "scene 490 alone" | cut -c1-300 | tee $O/scene_alone.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 200 python tools/fuzz_structural.py 6802 140 2>&1 | tail -2 | cut -c1-600 | tee $O/fuzz_structural_6802.txt
