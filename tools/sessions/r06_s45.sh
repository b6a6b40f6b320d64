# round 6, session 40: compiler-flag variants of the three bench units (tools/experiments/variants/make_quick_variant.sh), one process per library, two rounds
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s45
mkdir -p $O
V=$GRAFT_REPO_ROOT/tools/experiments/variants
for round in 1 2; do
for scene in ragdoll crowd; do
  for lib in "" slp2 slp5 slp10 os; do
    if [ -z "$lib" ]; then unset BEPUHIP_LIB; label="product"; else export BEPUHIP_LIB=$V/libbepuhip_$lib.so; label=$lib; fi
    echo -n "$scene $label: "; STEPS=300 timeout 300 python tools/ab_scene.py $scene "x:" 2>&1 | grep "ms/step" | cut -c1-110
  done
done
done 2>&1 | tee $O/ab_compiler_flags.txt
