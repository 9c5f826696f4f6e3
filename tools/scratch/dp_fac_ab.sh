#!/bin/bash
# DP code path at one rank (RCCL group of 1): gathered G.0 factors on / off, and the single-process path, interleaved
export MASTER_ADDR=127.0.0.1
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for r in 1 2; do
  run single
  MASTER_PORT=2961$r RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 RNAGAN_DP_G0_FACTORS=0 run dp_allreduce_everything
  MASTER_PORT=2962$r RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 RNAGAN_DP_G0_FACTORS=1 run dp_g0_factors
done
