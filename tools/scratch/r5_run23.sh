#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_23; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "f32_matrix" > $O/pytest_default.log 2>&1; tail -3 $O/pytest_default.log | cut -c1-250
RNAGAN_F32MMA=2 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log | cut -c1-250
RNAGAN_F32MMA=2 timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "fp32 or full_size" > $O/pytest_train.log 2>&1; tail -3 $O/pytest_train.log | cut -c1-250
for r in 1 2 3; do for v in 1 2; do
  RNAGAN_F32MMA=$v timeout 600 python3 bench.py --gpus 1 --precision fp32 --steps 6 --warmup 12 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*' | sed "s/^/f32mma=$v /"
done; done
