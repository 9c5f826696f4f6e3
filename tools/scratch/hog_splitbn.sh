#!/bin/bash
# fused split-K BatchNorm kernels (grid-wide rendezvous) on / off while a "collective" holds CUs (see hog_ab.sh)
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
p=29900
for r in 1 2; do
  for hog in "" "32,500" "-32,500"; do for f in 0 1; do
    p=$((p+1)); export MASTER_PORT=$p
    RNAGAN_DEBUG_HOG=$hog RNAGAN_SPLIT_BN_DP=$f run "hog=[$hog] fused_split_bn=$f"
  done; done
done
