#!/bin/bash
# round 5, call 2: BatchNorm row kernels with raw loads (RawVec) -- parity + build A/B + bn microbench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_2
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "bn or batchnorm or BatchNorm or tangent or double" > gpurun_out/r5_2/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_2/pytest.log
tail -5 gpurun_out/r5_2/pytest.log
for v in base rawvec; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  echo "== $v" >> gpurun_out/r5_2/bench_bn.log
  timeout 300 python tools/bench_bn.py >> gpurun_out/r5_2/bench_bn.log 2>&1
done
cat gpurun_out/r5_2/bench_bn.log
bash tools/scratch/ab_step.sh base rawvec base rawvec > gpurun_out/r5_2/ab_libs.log 2>&1
cat gpurun_out/r5_2/ab_libs.log
