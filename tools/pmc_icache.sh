#!/bin/bash
# Instruction-cache counters of cluster_kernel on the GPU box (developer helper).
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "InstrFetchLatency"; do
  rm -rf /tmp/pmc_ic
  STEPS=3 WARM=2 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_ic -o out -- python $GRAFT_REPO_ROOT/tools/perf_cluster.py clusters > /tmp/pmc_ic.log 2>&1
  f=$(find /tmp/pmc_ic -name "*counter_collection.csv" | head -1)
  echo "== $set"; [ -z "$f" ] && { tail -5 /tmp/pmc_ic.log; continue; }
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "cluster_kernel" not in k: continue
    agg["cluster_kernel"][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k, d in agg.items():
    print("  ", k, {c: round(v / n[c]) for c, v in d.items()})
PY
done
