#!/bin/bash
# round 5, call 1: image-side rows128 kernels -- parity, microbenchmark, interleaved step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_1
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "image_side or input_gradient or first_down or skinny or last_up or rows128" > gpurun_out/r5_1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_1/pytest.log
tail -5 gpurun_out/r5_1/pytest.log
timeout 300 python tools/bench_skinny.py > gpurun_out/r5_1/bench_skinny.log 2>&1
cat gpurun_out/r5_1/bench_skinny.log
timeout 900 python tools/ab_step.py --variants "general:skinny128=0;rows128:skinny128=1" --rounds 5 --steps 20 --json gpurun_out/r5_1/ab_skinny.json > gpurun_out/r5_1/ab_step.log 2>&1
tail -12 gpurun_out/r5_1/ab_step.log
