#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_18; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "wire or one_launch or slabs_inside" > $O/pytest_op.log 2>&1; tail -5 $O/pytest_op.log
timeout 1500 python -m pytest tests/test_dp_gpu.py tests/test_dp2_gpu.py -x -q -m gpu > $O/pytest_dp.log 2>&1; tail -8 $O/pytest_dp.log | cut -c1-300
bash tools/scratch/r5_dp1.sh > $O/dp1.log 2>&1; cat $O/dp1.log
