#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_run1_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run1_tests.log
tail -5 gpurun_out/r4_run1_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_run1_bench.json 2> gpurun_out/r4_run1_bench.err
tail -25 gpurun_out/r4_run1_bench.err
cat gpurun_out/r4_run1_bench.json | cut -c1-1500
