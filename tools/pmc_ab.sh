#!/bin/bash
# A/B hardware-counter comparison of two builds of libbepuhip on the GPU box (developer helper).
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pmc_$v
    BEPUHIP_LIB=$GRAFT_REPO_ROOT/bepuphysics2_amd/csrc/libbepuhip_$v.so STEPS=3 WARM=2 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$v -o out -- python $GRAFT_REPO_ROOT/tools/perf_cluster.py clusters > /tmp/pmc_$v.log 2>&1
    f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
    echo "== $v: $set"; [ -z "$f" ] && { tail -5 /tmp/pmc_$v.log; find /tmp/pmc_$v | head; continue; }
    python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    k = ("cluster_kernel" + k.split("cluster_kernel")[1][:12]) if "cluster_kernel" in k else k[:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "cluster_kernel" in k:
        print("  ", k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
  done
done
