#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_26; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "image_side or skinny or first_down or last_up" > $O/pytest_op.log 2>&1; tail -4 $O/pytest_op.log | cut -c1-250
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py tests/test_bench_step_gpu.py tests/test_dp_gpu.py tests/test_dp2_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; tail -4 $O/pytest_train.log | cut -c1-250
timeout 900 python tools/ab_step.py --variants "on:RNAGAN_SKINNY_BIAS=1;off:RNAGAN_SKINNY_BIAS=0;on2:RNAGAN_SKINNY_BIAS=1" --rounds 5 --steps 40 --json $O/ab_skinny_bias.json > $O/ab.log 2>&1; tail -5 $O/ab.log | cut -c1-250
