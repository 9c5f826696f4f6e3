#!/bin/bash
# one-rank data-parallel route against the single-process route, with and without the wire-direct weight gradients (separate
# processes, alternating, three rounds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; ms=$(env "$@" python3 bench.py --gpus 1 --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*'); echo "$tag: $ms"; }
for r in 1 2 3; do
run single X=1
run dp1_wire1 RNAGAN_FORCE_DP=1 RNAGAN_DP_WIRE_DIRECT=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$r
run dp1_wire0 RNAGAN_FORCE_DP=1 RNAGAN_DP_WIRE_DIRECT=0 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2962$r
done
