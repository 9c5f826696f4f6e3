#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_31; mkdir -p $O
timeout 1500 python tools/ab_step.py --variants "base:;a2k:bn_apply_blocks=2048;a8k:bn_apply_blocks=8192;a16k:bn_apply_blocks=16384,bn_apply_cap=16384;r768:bn_reduce_blocks=768;r1024:bn_reduce_blocks=1024" --rounds 4 --steps 40 --json $O/ab.json > $O/ab.log 2>&1; tail -9 $O/ab.log | cut -c1-200
