#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_7; mkdir -p $O
timeout 300 python tools/scratch/dbg_convp64.py 2>&1 | grep mismatch > $O/dbg.log; cat $O/dbg.log
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -s > $O/pytest_ops.log 2>&1
echo "rc=$?" >> $O/pytest_ops.log; grep -E "nsplit|passed|failed|rc=" $O/pytest_ops.log | tail -12
timeout 900 python tools/ab_step.py --variants "reduce:losses.SLAB_ADAM=0;slabadam:losses.SLAB_ADAM=1" --rounds 5 --steps 20 --json $O/ab_slabadam.json > $O/ab_step.log 2>&1; tail -4 $O/ab_step.log
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_bench_step_gpu.py tests/test_engine_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1
echo "rc=$?" >> $O/pytest_train.log; tail -3 $O/pytest_train.log
timeout 1200 python -m pytest tests/test_dp2_gpu.py -x -q -m gpu -k "world4 or world8" -s > $O/pytest_dp.log 2>&1
echo "rc=$?" >> $O/pytest_dp.log; grep -E "passed|failed|rc=|wire" $O/pytest_dp.log | tail -5
