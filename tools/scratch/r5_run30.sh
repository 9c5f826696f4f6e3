#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_30; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_layer0 or g0" > $O/pytest_op.log 2>&1; tail -4 $O/pytest_op.log | cut -c1-250
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py tests/test_bench_step_gpu.py tests/test_dp_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; tail -4 $O/pytest_train.log | cut -c1-250
for v in 1 0; do
  export RNAGAN_G0_PACK_IN_STEP=$v
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/prof$v.json 2> $O/prof$v.err
  python3 tools/prof_summary.py $O/prof 60 > $O/prof$v.txt
  rm -rf $O/prof
  echo "== pack in step $v"; head -1 $O/prof$v.txt; grep -n "g0_wgrad_adam\|transpose_bf16" $O/prof$v.txt | cut -c1-170
done
unset RNAGAN_G0_PACK_IN_STEP
timeout 900 python tools/ab_step.py --variants "on:RNAGAN_G0_PACK_IN_STEP=1;off:RNAGAN_G0_PACK_IN_STEP=0" --rounds 5 --steps 40 --json $O/ab.json > $O/ab.log 2>&1; tail -4 $O/ab.log | cut -c1-200
