# round 6, session 53: what the 11 us between two back-to-back solves depend on — scene size (1,500 / 15,000 ragdolls: ten times fewer dirty lines at the kernel's end),
# plan kind (pile: cooperative launch, a unit without scratch)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s53
mkdir -p $O
for spec in "ragdoll 1500" "ragdoll 15000" "pile 0"; do
  set -- $spec
  RAGDOLLS=$2 STEPS=150 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o r -- python $GRAFT_REPO_ROOT/tools/ab_scene.py $1 "x:" > $O/ab_$1_$2.txt 2>&1
  T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
  echo "== $spec"; grep "ms/step" $O/ab_$1_$2.txt | cut -c1-100; python $GRAFT_REPO_ROOT/tools/trace_durations.py $T cluster_kernel 100 | sed -n 2,5p
  rm -rf $O/prof
done | tee $O/gaps_by_scene.txt
