#!/bin/bash
# interleaved A/B/C of an environment variable's values on one box: ab_env3.sh VAR v1 v2 [v3]
V=$1; shift
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for r in 1 2 3; do for val in "$@"; do export $V=$val; run "$V=$val"; done; done
