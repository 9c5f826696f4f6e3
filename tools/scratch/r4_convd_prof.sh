#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
rm -rf gpurun_out/r4_cd_prof
RNAGAN_CONVD=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_cd_prof -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_cd_prof.json 2> gpurun_out/r4_cd_prof.err
python3 tools/prof_groups.py gpurun_out/r4_cd_prof "" 400 > gpurun_out/r4_cd_groups_$v.txt 2>&1
done
rm -rf gpurun_out/r4_cd_prof
