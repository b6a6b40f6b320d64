# round 6, session 66: a context alone on its device launches its split plans plainly: GPU suite, pile / crowd with default flags (tools/ab_scene.py), forced cooperative beside it
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s66
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
for scene in pile crowd; do STEPS=300 timeout 300 python tools/ab_scene.py $scene "default flags:" "forced cooperative:BEPUHIP_COOPERATIVE=1" "default flags again:" 2>&1 | grep "ms/step\|bodies," | cut -c1-120; done | tee $O/ab_split_default_launch.txt
timeout 300 python tools/soak.py 3 120 2>&1 | tail -2 | tee $O/soak.txt
