#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for prec in fp32 bf16; do
  PREC=$prec timeout 300 python3 tools/scratch/dbg_enc200.py > gpurun_out/r4_dbg_enc200_$prec.log 2>&1; echo "rc $?" >> gpurun_out/r4_dbg_enc200_$prec.log
  tail -4 gpurun_out/r4_dbg_enc200_$prec.log
done
GRAPHS=1 PREC=bf16 SIZE=256 N=64 timeout 300 python3 tools/scratch/dbg_enc200.py > gpurun_out/r4_dbg_enc200_full.log 2>&1; echo "rc $?" >> gpurun_out/r4_dbg_enc200_full.log
tail -4 gpurun_out/r4_dbg_enc200_full.log
timeout 900 python3 -m pytest tests/test_quality_ab_gpu.py tests/test_trainer_gpu.py::test_cli_stock_wgan_literal_command tests/test_dp2_gpu.py -x -q -s > gpurun_out/r4_run2_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run2_tests.log
tail -15 gpurun_out/r4_run2_tests.log
timeout 1500 python3 tools/train_quality_ab.py --out gpurun_out/r4_quality_ab.json > gpurun_out/r4_quality_ab.md 2> gpurun_out/r4_quality_ab.err; echo "ab rc $?"
tail -20 gpurun_out/r4_quality_ab.err; cat gpurun_out/r4_quality_ab.md
