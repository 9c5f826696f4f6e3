#!/bin/bash
# bf16 weight-gradient slabs (wslab16): op tests, step A/B, per-tensor update cosines of the full-size iteration under both settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_13; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s -k "one_launch or slabs_inside" > $O/pytest_op.log 2>&1; grep -E "passed|failed|relative L2|Error|assert" $O/pytest_op.log | tail -14
timeout 900 python tools/ab_step.py --variants "w32:wslab16=0;w16:wslab16=1" --rounds 5 --steps 40 --json $O/ab_wslab16.json > $O/ab_wslab16.log 2>&1; tail -4 $O/ab_wslab16.log | cut -c1-300
for v in 0 1; do
  RNAGAN_WSLAB16=$v timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bench_step_gpu.py -x -q -m gpu -s -k "bench or batch64 or batch_64 or full_size" > $O/quality_$v.log 2>&1
  echo "== RNAGAN_WSLAB16=$v rc=$?"; grep -E "passed|failed|worst" $O/quality_$v.log | tail -6 | cut -c1-300
done
