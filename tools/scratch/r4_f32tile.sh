#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 0 1; do
  ms=$(RNAGAN_F32TILE=$v python3 bench.py --precision fp32 --steps 3 --warmup 6 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "f32tile=$v: fp32 $ms"
done; done
