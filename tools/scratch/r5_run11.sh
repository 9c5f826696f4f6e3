#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_11; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1
echo "rc=$?" >> $O/pytest_ops.log; tail -3 $O/pytest_ops.log
timeout 900 python tools/ab_step.py --variants "slab32:slab16=0;slab16:slab16=1" --rounds 5 --steps 20 --json $O/ab_slab16.json > $O/ab_step.log 2>&1; tail -4 $O/ab_step.log | cut -c1-250
for v in 0 1; do
  RNAGAN_SLAB16=$v timeout 900 python -m pytest tests/test_bench_step_gpu.py tests/test_train_gpu.py -x -q -m gpu -s -k "bench or batch_64 or full_size or unselected_inputs_loss" > $O/quality_$v.log 2>&1
  echo "== RNAGAN_SLAB16=$v rc=$?"; grep -E "passed|failed|cos|seeds x 3|rel|loss" $O/quality_$v.log | tail -12 | cut -c1-300
done
