#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_groups.py gpurun_out/r4_fp32_prof Wgrad 12 --seq 16 > gpurun_out/r4_fp32_groups_seq.txt 2>&1
rm -rf gpurun_out/r4_fp32_prof
