#!/bin/bash
# round-end evidence: the driver's bench command, then its rocprofv3 kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
tail -4 gpurun_out/r3_final_bench.err
rm -rf gpurun_out/r3_final_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_final_prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3_final_prof.json 2> gpurun_out/r3_final_prof.err
python3 tools/prof_summary.py gpurun_out/r3_final_prof 60 > gpurun_out/r3_final_prof.txt
head -12 gpurun_out/r3_final_prof.txt
cp $(ls gpurun_out/r3_final_prof/*/*kernel_stats.csv | head -1) gpurun_out/r3_final_kernel_stats.csv
rm -rf gpurun_out/r3_final_prof   # raw traces stay on the box (64 MiB copy-back limit)
