#!/bin/bash
# fp32 mode on the bf16 matrix cores (f32mma = 2: each fp32 product as six bf16 products): tests under the option, fp32 step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_19; mkdir -p $O
RNAGAN_F32MMA=2 timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1; tail -5 $O/pytest_ops.log | cut -c1-250
RNAGAN_F32MMA=2 timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_vae_gpu.py -x -q -m gpu -s > $O/pytest_train.log 2>&1; grep -E "passed|failed|fp32" $O/pytest_train.log | tail -12 | cut -c1-250
for r in 1 2; do for v in 1 2; do
  RNAGAN_F32MMA=$v timeout 600 python3 bench.py --gpus 1 --precision fp32 --steps 6 --warmup 12 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*' | sed "s/^/f32mma=$v /"
done; done
