# round 6, session 72: the failing structural scene alone (generator state saved in front of it), under switches that take parts of the path out
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s72
mkdir -p $O
run() { echo "== $1"; shift; env "$@" timeout 100 python tools/probes/replay_fuzz_structural.py 6802 490 --load tests/golden/fuzz_structural_6802_490_state.json 2>&1 | grep -v "^\[W\|amdgpu.ids" | grep "scene 490 alone\|bepuhip" | cut -c1-260 | tail -12; }
(run "as shipped" X=1
run "updates leave the split plan (BEPUHIP_NO_SPLIT_SOFT_UPDATES=1)" BEPUHIP_NO_SPLIT_SOFT_UPDATES=1
run "one planner thread" BEPUHIP_PLAN_THREADS=1
run "no local hand-offs" BEPUHIP_SPLIT_LOCAL_HANDOFF=0
run "no merged manifold items" BEPUHIP_SPLIT_FUSE=0
run "plan stats" BEPUHIP_PLAN_STATS=1) 2>&1 | tee $O/variants.txt
