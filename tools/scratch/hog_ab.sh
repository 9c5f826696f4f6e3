#!/bin/bash
# what does sharing CUs with a collective cost the overlapped compute?  One-rank DP path (RCCL group of 1) with a kernel that holds
# B CUs for T microseconds wherever an all-reduce runs (RNAGAN_DEBUG_HOG), overlap on / off
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
p=29800
for r in 1 2; do
  for spec in "" "16,500" "32,500" "64,500" "32,1000"; do
    p=$((p+1)); export MASTER_PORT=$p
    RNAGAN_DEBUG_HOG=$spec RNAGAN_DP_OVERLAP=1 run "overlap    hog=[$spec]"
  done
  p=$((p+1)); export MASTER_PORT=$p
  RNAGAN_DEBUG_HOG="32,500" RNAGAN_DP_OVERLAP=0 run "no-overlap hog=[32,500]"
done
