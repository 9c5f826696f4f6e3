#!/bin/bash
# kernel statistics of the benchmark under two settings of one RNAGAN_* knob: prof_ab.sh KNOB A B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $2 $3; do
  export $1=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1_$v -- python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 > gpurun_out/prof_$1_$v.json 2> gpurun_out/prof_$1_$v.err
  python tools/prof_summary.py gpurun_out/prof_$1_$v 45 > gpurun_out/prof_$1_$v.txt
done
