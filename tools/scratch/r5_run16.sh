#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_16; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "batchnorm or bn" > $O/pytest_op.log 2>&1; tail -3 $O/pytest_op.log
timeout 1800 python tools/ab_step.py --variants "r0:bn_rev=0;r3:bn_rev=3;r4:bn_rev=4;r5:bn_rev=5;r6:bn_rev=6;r7:bn_rev=7" --rounds 5 --steps 40 --json $O/ab_bn_rev.json > $O/ab_bn_rev.log 2>&1; tail -8 $O/ab_bn_rev.log | cut -c1-200
