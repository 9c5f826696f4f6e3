#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_run3_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run3_tests.log
tail -8 gpurun_out/r4_run3_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_run3_bench.json 2> gpurun_out/r4_run3_bench.err
grep -E "timed region|extra fp32|extra enc200|FAILED|cpu baseline:" gpurun_out/r4_run3_bench.err | cut -c1-1800
RNAGAN_F32MMA=0 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r4_run3_bench_valu.json 2> gpurun_out/r4_run3_bench_valu.err
grep -E "extra fp32" gpurun_out/r4_run3_bench_valu.err | cut -c1-600
