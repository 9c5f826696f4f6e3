#!/bin/bash
# A/B of builds of the fp32 matrix-core kernel on one box: step time of bench.py --precision fp32, two interleaved rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
python3 -m pytest tests/test_ops_gpu.py -q -x -k "conv_down_up_wgrad or f32_matrix or g0_and_head or linear or upconv3" > gpurun_out/r4_f32ab_tests.log 2>&1; echo "ops tests (default build) rc $?"; tail -2 gpurun_out/r4_f32ab_tests.log
for r in 1 2; do for v in "$@"; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  ms=$(python3 bench.py --precision fp32 --steps 3 --warmup 6 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "$v: fp32 $ms"
done; done
cp tools/scratch/lib_f32bk32.so rna_gan_amd/librnagan_hip.so
python3 -m pytest tests/test_ops_gpu.py -q -x -k "conv_down_up_wgrad or f32_matrix or g0_and_head or linear" > gpurun_out/r4_f32ab_tests_bk32.log 2>&1; echo "ops tests (bk32 build) rc $?"; tail -2 gpurun_out/r4_f32ab_tests_bk32.log
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
python3 -m pytest tests/test_train_gpu.py -q -s -k unselected > gpurun_out/r4_seedstats.log 2>&1; grep -E "fp32:|bf16:|passed|failed" gpurun_out/r4_seedstats.log
