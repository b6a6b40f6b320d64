#!/bin/bash
# round 4, session 5: the XCD questions (workgroup -> XCD mapping; record hand-offs at workgroup scope inside an XCD)
set -u
O=gpurun_out/r04_s5; mkdir -p $O
timeout 300 tools/probes/xcd_scope_probe.bin 2>&1 | tee $O/xcd_scope_probe.txt
