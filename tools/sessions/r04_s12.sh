#!/bin/bash
# round 4, session 12: what WRITE_SIZE counts for the record stores (calibration probe), waves per cluster on the split plans, the pile's per-item timeline
set -u
O=gpurun_out/r04_s12; mkdir -p $O
export TMPDIR=/tmp
for counter in WRITE_SIZE FETCH_SIZE; do
  ( cd /tmp && rm -rf /tmp/wsp_$counter && timeout 300 rocprofv3 --pmc $counter --output-format csv -d /tmp/wsp_$counter -o p -- $GRAFT_REPO_ROOT/tools/probes/write_size_probe.bin ) > $O/probe_$counter.log 2>&1
  python - $counter <<'PY' | tee -a $O/write_size_probe.txt
import csv, glob, sys
counter = sys.argv[1]
for f in glob.glob(f"/tmp/wsp_{counter}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            print(f"{counter} {r['Kernel_Name'][:60]:60s} {float(r['Counter_Value']) * 1024 / 1e6:10.2f} MB (counter in KiB)")
PY
done
tail -1 $O/probe_WRITE_SIZE.log | tee -a $O/write_size_probe.txt
for scene in pile crowd; do
  BEPUHIP_ROW_POLICY=0 timeout 400 python tools/ab_scene.py $scene "512 threads:" "768 threads:BEPUHIP_SPLIT_THREADS=768" "1024 threads:BEPUHIP_SPLIT_THREADS=1024" 2>&1 | grep "ms/step\|bodies" | tee -a $O/split_threads.txt
done
SCENE=pile timeout 300 python tools/cluster_trace.py 2>&1 | head -60 > $O/pile_trace.txt; head -16 $O/pile_trace.txt
