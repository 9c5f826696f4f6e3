#!/bin/bash
# under an emulated collective (32 CUs x 500 us): grids below the CU count for the split-K conv / weight-gradient launches?
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 RNAGAN_DEBUG_HOG="-32,500"
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
p=30000
for r in 1 2; do
  for b in 256 224 192; do
    p=$((p+1)); export MASTER_PORT=$p
    RNAGAN_CONV8_BLOCKS=$b RNAGAN_WGRAD8_BLOCKS=$b run "blocks=$b"
  done
done
