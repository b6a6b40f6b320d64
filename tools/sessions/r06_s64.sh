# round 6, session 64: the launch-per-batch script of session 63 without the profiler (it died with SIGSEGV under rocprofv3), then under rocprofv3 with BEPUHIP no-graph
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s64
mkdir -p $O
sed -n '/^cat > \/tmp\/lpb.py/,/^PY$/p' $GRAFT_REPO_ROOT/tools/sessions/r06_s63.sh | sed '1d;$d' > /tmp/lpb.py
echo "== no profiler"; timeout 300 python /tmp/lpb.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | grep -v "^    @" | tail -5
echo "== rocprofv3, graphs off (use_graph=False)"; sed -i 's/HipSolver(use_clusters=False)/HipSolver(use_clusters=False, use_graph=False)/' /tmp/lpb.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python /tmp/lpb.py 2>&1 | grep -v "^\[W\|^W2\|^E2\|amdgpu.ids" | grep -v "^    @" | tail -5
ls $O/prof 2>/dev/null | head -3
