#!/bin/bash
set -u
O=gpurun_out/r03_s3; mkdir -p $O
export TMPDIR=/tmp BEPUHIP_PREFETCH=0
timeout 600 python tools/ab_scene.py pile "default:" "cover:BEPUHIP_SPLIT_COVER=1" "cover+refine2:BEPUHIP_SPLIT_COVER=1,BEPUHIP_SPLIT_REFINE=2" "refine2:BEPUHIP_SPLIT_REFINE=2" "cover,plain-launch:BEPUHIP_SPLIT_COVER=1,BEPUHIP_COOPERATIVE=0" "poll0:BEPUHIP_SHARED_POLL=0" "threads768:BEPUHIP_SPLIT_THREADS=768" 2>&1 | tee $O/ab_pile_cut.txt
timeout 600 python tools/ab_scene.py crowd "default:" "cover:BEPUHIP_SPLIT_COVER=1" "cover+refine2:BEPUHIP_SPLIT_COVER=1,BEPUHIP_SPLIT_REFINE=2" "cover,plain-launch:BEPUHIP_SPLIT_COVER=1,BEPUHIP_COOPERATIVE=0" 2>&1 | tee $O/ab_crowd_cut.txt
BEPUHIP_PLAN_STATS=1 BEPUHIP_SPLIT_COVER=1 STEPS=5 timeout 300 python tools/ab_scene.py crowd "cover:" 2>&1 | grep -i "split plan\|row policy" | tee $O/plan_stats.txt
rocm-smi --showclocks 2>&1 | head -3
