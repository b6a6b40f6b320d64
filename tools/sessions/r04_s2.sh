#!/bin/bash
# round 4, session 2: statistics on the nondeterministic mismatch of fuzz seed 81, ordinal 91 (library as committed at the end of round 3)
set -u
O=gpurun_out/r04_s2; mkdir -p $O
export TMPDIR=/tmp
timeout 700 python tools/race_hunt.py 81 91 200 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $O/race_hunt.txt
