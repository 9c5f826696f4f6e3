#!/bin/bash
# interleaved A/B of an environment switch on one box: ab_env.sh VAR  (VAR=0 vs VAR=1), step time of the benchmark
V=$1
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['losses_last_step'])"; }
for r in 1 2 3; do
  env $V=0 python3 -c "pass"; export $V=0; run "$V=0"
  export $V=1; run "$V=1"
done
