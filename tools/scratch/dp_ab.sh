#!/bin/bash
# step time of the benchmark: single-process path vs the data-parallel code path with ONE rank (RCCL group of 1), with the
# D-loss prefix = D(real) forward only (round 2) / forward + backward (round 3); interleaved, two rounds
export MASTER_ADDR=127.0.0.1
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for r in 1 2; do
  run single
  MASTER_PORT=2951$r RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 RNAGAN_DP_PREFIX_BWD=0 run dp_prefix_fwd
  MASTER_PORT=2952$r RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 RNAGAN_DP_PREFIX_BWD=1 run dp_prefix_fwd_bwd
done
