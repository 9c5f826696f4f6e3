#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_17; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest_op.log 2>&1; tail -3 $O/pytest_op.log
timeout 1800 python tools/ab_step.py --variants "base:bn_rev=4;c1:conv_rev=1;c2:conv_rev=2;c3:conv_rev=3;c3f:conv_rev=3,bn_rev=6;s7:bn_swz=7;s7c3:bn_swz=7,conv_rev=3" --rounds 4 --steps 40 --json $O/ab_rev.json > $O/ab_rev.log 2>&1; tail -9 $O/ab_rev.log | cut -c1-200
