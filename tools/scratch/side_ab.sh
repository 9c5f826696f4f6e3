#!/bin/bash
# weight-gradient launches on a forked side stream inside the step graphs (RNAGAN_SIDE_STREAM=1) vs the serial chain
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>gpurun_out/side_$1.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['losses_last_step'])"; }
for r in 1 2; do
  run serial
  RNAGAN_SIDE_STREAM=1 run side
done
tail -3 gpurun_out/side_side.err
