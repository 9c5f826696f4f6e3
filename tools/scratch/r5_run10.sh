#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_10; mkdir -p $O
timeout 1500 python tools/tolerance_stats.py --seeds 200 --sizes 32,64 --out $O/tolerance_statistics.txt > $O/tol.log 2>&1; cat $O/tolerance_statistics.txt
bash tools/scratch/ab_step.sh head head2 > $O/ab_libs.log 2>&1; cat $O/ab_libs.log
