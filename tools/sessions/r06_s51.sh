# round 6, session 51: buffer_wbl2 at the end of every cluster (BEPU_VARIANT_END_WRITEBACK) against the product library, generic hot unit, one process per library, two rounds
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s51
mkdir -p $O
V=$GRAFT_REPO_ROOT/tools/experiments/variants
for round in 1 2; do
  for lib in "" wbl2; do
    if [ -z "$lib" ]; then unset BEPUHIP_LIB; label="product"; else export BEPUHIP_LIB=$V/libbepuhip_$lib.so; label=$lib; fi
    echo -n "ragdoll $label: "; STEPS=300 timeout 300 python tools/ab_scene.py ragdoll "x:" 2>&1 | grep "ms/step" | cut -c1-110
  done
done 2>&1 | tee $O/ab_end_writeback.txt
