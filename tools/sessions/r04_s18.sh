#!/bin/bash
# round 4, session 18: what FETCH_SIZE counts for the polls (lone / paired 16-byte agent-scope loads)
set -u
O=gpurun_out/r04_s18; mkdir -p $O
export TMPDIR=/tmp
for counter in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/wsp_$counter && timeout 300 rocprofv3 --pmc $counter --output-format csv -d /tmp/wsp_$counter -o p -- $GRAFT_REPO_ROOT/tools/probes/write_size_probe.bin ) > $O/probe_$counter.log 2>&1
  python - $counter <<'PY' | tee -a $O/access_size_probe.txt
import csv, glob, sys
counter = sys.argv[1]
for f in glob.glob(f"/tmp/wsp_{counter}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            print(f"{counter} {r['Kernel_Name'][:60]:60s} {float(r['Counter_Value']) * 1024 / 1e6:10.2f} MB (counter in KiB, as reported)")
PY
done
tail -2 $O/probe_FETCH_SIZE.log | tee -a $O/access_size_probe.txt
