# Which of the pool's two box classes is this, and what differs: the bench scene with the row policy pinned to plain, the two probes, the instruction-cache counters.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/boxclass
T=gpurun_out/boxclass/tmp.txt
(echo "== bench scene, plain row policy / non-temporal row policy"; BEPUHIP_ROW_POLICY=0 STEPS=200 WARM=60 timeout 200 python tools/perf_cluster.py clusters 2>&1 | tail -1;
 BEPUHIP_ROW_POLICY=1 STEPS=200 WARM=60 timeout 200 python tools/perf_cluster.py clusters 2>&1 | tail -1;
 echo "== tools/probes/clock_probe.bin"; timeout 120 tools/probes/clock_probe.bin; echo "== tools/probes/icache_probe.bin"; timeout 120 tools/probes/icache_probe.bin;
 echo "== tools/pmc_icache.sh (cluster_kernel, plain row policy)"; BEPUHIP_ROW_POLICY=0 timeout 600 bash tools/pmc_icache.sh 2>&1 | tail -6) > $T 2>&1
if sed -n 2p $T | grep -q "0\.2[0-9][0-9] ms"; then mv $T gpurun_out/boxclass/slow.txt; echo SLOW; else mv $T gpurun_out/boxclass/fast.txt; echo FAST; fi
