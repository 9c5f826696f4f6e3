#!/bin/bash
# (b2): weight gradient + Adam step in one launch for the no-split layers -- op test, the train / quality tests, the step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_12; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "one_launch or slabs_inside" > $O/pytest_op.log 2>&1; tail -5 $O/pytest_op.log
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_quality_ab_gpu.py tests/test_engine_gpu.py tests/test_dp_gpu.py -q -m gpu > $O/pytest_train.log 2>&1; tail -8 $O/pytest_train.log
timeout 900 python tools/ab_step.py --variants "off:RNAGAN_WGRAD_ADAM=0;on:RNAGAN_WGRAD_ADAM=1" --rounds 5 --steps 40 --json $O/ab_wgrad_adam.json > $O/ab_wgrad_adam.log 2>&1; tail -6 $O/ab_wgrad_adam.log | cut -c1-300
