# round 6, session 58: one buffer_wbl2 per cluster (one wave, behind the body write-back) against the product library, generic hot unit, one process per library, two rounds
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s58
mkdir -p $O
V=$GRAFT_REPO_ROOT/tools/experiments/variants
for round in 1 2; do
  for lib in "" wbl2one; do
    if [ -z "$lib" ]; then unset BEPUHIP_LIB; label="product"; else export BEPUHIP_LIB=$V/libbepuhip_$lib.so; label=$lib; fi
    echo -n "ragdoll $label: "; STEPS=300 timeout 300 python tools/ab_scene.py ragdoll "x:" 2>&1 | grep "ms/step" | cut -c1-110
  done
done 2>&1 | tee $O/ab_end_writeback_one_wave.txt
