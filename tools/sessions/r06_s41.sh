# round 6, session 41: the library compiled without the SLP vectoriser (-fno-slp-vectorize): GPU suite, bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s41
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json
