#!/bin/bash
# the tolerance statistics again on the final tree (slab16, wslab16 and the image-side by-products came after the first collection),
# and under the opt-in fp32 form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_33; mkdir -p $O
timeout 2400 python tools/tolerance_stats.py --seeds 200 --out $O/tolerance_final.txt > $O/stats.log 2>&1; tail -5 $O/stats.log | cut -c1-400
RNAGAN_F32MMA=2 timeout 1500 python tools/tolerance_stats.py --seeds 100 --sizes 64 --out $O/tolerance_f32mma2.txt > $O/stats2.log 2>&1; head -1 $O/tolerance_f32mma2.txt | cut -c1-400
