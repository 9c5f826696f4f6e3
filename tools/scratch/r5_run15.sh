#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_15; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "batchnorm or bn or one_launch or slabs_inside" > $O/pytest_op.log 2>&1; tail -3 $O/pytest_op.log
timeout 1200 python tools/ab_step.py --variants "r0:bn_rev=0;r1:bn_rev=1;r2:bn_rev=2;r3:bn_rev=3" --rounds 5 --steps 40 --json $O/ab_bn_rev.json > $O/ab_bn_rev.log 2>&1; tail -6 $O/ab_bn_rev.log | cut -c1-200
