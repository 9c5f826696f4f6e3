#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_9; mkdir -p $O
timeout 1200 python tools/ab_step.py --variants "all:;noskinny:skinny128=0;noslab:losses.SLAB_ADAM=0" --rounds 4 --steps 20 --json $O/ab_opts.json > $O/ab_opts.log 2>&1; tail -5 $O/ab_opts.log | cut -c1-200
bash tools/scratch/ab_step.sh head old_convp old_conv8 old_wgrad8 > $O/ab_libs.log 2>&1; cat $O/ab_libs.log
