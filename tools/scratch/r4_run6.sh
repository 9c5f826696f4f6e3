#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_run6_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run6_tests.log
tail -6 gpurun_out/r4_run6_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_run6_bench.json 2> gpurun_out/r4_run6_bench.err
grep -E "timed region|extra fp32|FAILED" gpurun_out/r4_run6_bench.err | cut -c1-1500
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_summary.py gpurun_out/r4_fp32_prof 40 > gpurun_out/r4_fp32_prof6.txt 2>&1
head -24 gpurun_out/r4_fp32_prof6.txt | cut -c1-200
rm -rf gpurun_out/r4_fp32_prof
