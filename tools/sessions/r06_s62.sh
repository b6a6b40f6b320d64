# round 6, session 62: the round's last check of the committed tree: build(), smoke(), GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s62
mkdir -p $O
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke OK')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; wc -l $O/bench.json; cut -c1-250 $O/bench.json
