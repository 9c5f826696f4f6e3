#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_summary.py gpurun_out/r4_fp32_prof 40 > gpurun_out/r4_fp32_prof8.txt 2>&1
head -34 gpurun_out/r4_fp32_prof8.txt | cut -c1-190
rm -rf gpurun_out/r4_fp32_prof
