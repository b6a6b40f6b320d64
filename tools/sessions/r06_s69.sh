# round 6, session 69: the structural fuzzer's mismatch of session 68 (seed 6802, scene 490: bodies exact, impulse / prestep rows not) replayed on the shipped library and on a
# library without today's two functional changes (tail workgroups always launched, the event pair around every solve)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s69
mkdir -p $O
(timeout 400 python tools/probes/replay_fuzz_structural.py 6802 492 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3 | cut -c1-700) | tee $O/replay_product.txt
(BEPUHIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/variants/libbepuhip_before_today.so timeout 400 python tools/probes/replay_fuzz_structural.py 6802 492 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3 | cut -c1-700) | tee $O/replay_before_today.txt
