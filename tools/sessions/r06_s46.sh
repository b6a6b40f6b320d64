# round 6, session 46: -Xarch_device -fno-slp-vectorize (the host half keeps its SLP): upload timing, GPU suite, bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s46
mkdir -p $O
timeout 300 python tools/perf_upload.py 2>&1 | tail -6 | tee $O/perf_upload.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --full-report $O/bench_full.json > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/s46/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["boundary"]["end_constraints_ms"], d["connected_scenes"]["pile_100k"]["ms_per_step"], d["connected_scenes"]["ragdoll_crowd"]["ms_per_step"])
PY
