#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_14; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s -k "one_launch or slabs_inside" > $O/pytest_op.log 2>&1; grep -E "passed|failed|relative L2|update cosine|Error|assert" $O/pytest_op.log | tail -14
for v in 0 1; do
  RNAGAN_WSLAB16=$v timeout 900 python -m pytest tests/test_bench_step_gpu.py -x -q -m gpu -s > $O/quality_$v.log 2>&1
  echo "== RNAGAN_WSLAB16=$v rc=$?"; grep -E "passed|failed|worst" $O/quality_$v.log | tail -6 | cut -c1-300
done
