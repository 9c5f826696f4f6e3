#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_25; mkdir -p $O
timeout 900 python tools/ab_step.py --variants "on:losses.SKINNY_SLAB_ADAM=1;off:losses.SKINNY_SLAB_ADAM=0;on2:losses.SKINNY_SLAB_ADAM=1" --rounds 5 --steps 40 --json $O/ab_skinny_slab.json > $O/ab.log 2>&1; tail -5 $O/ab.log | cut -c1-250
