#!/bin/bash
# kernel statistics of the Trainer.train loop (device-resident batches) for comparison with the benchmark's loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PREFETCH_AB_ONLY=resident
rm -rf gpurun_out/trainer_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trainer_prof -- python3 tools/scratch/prefetch_ab.py > gpurun_out/trainer_prof.log 2>&1
python3 tools/prof_summary.py gpurun_out/trainer_prof 45 > gpurun_out/trainer_prof.txt
cp $(ls gpurun_out/trainer_prof/*/*kernel_stats.csv | head -1) gpurun_out/trainer_kernel_stats.csv
rm -rf gpurun_out/trainer_prof
grep resident gpurun_out/trainer_prof.log
