#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_last_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_last_tests.log
tail -3 gpurun_out/r4_last_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_last_bench.json 2> gpurun_out/r4_last_bench.err
grep -E "timed region|fp32_step" gpurun_out/r4_last_bench.err | cut -c1-300
