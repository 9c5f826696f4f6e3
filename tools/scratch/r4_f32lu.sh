#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -q -x -k "image_side" 2>&1 | tail -3
python3 -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py -q -x -k "fp32 or float32 or reference_trainops or public_functional or tight or l2" 2>&1 | tail -3
for r in 1 2; do for v in 1 0; do
  ms=$(RNAGAN_LASTUP_WHOLE_=$v python3 -c "
import os,subprocess,sys
e=dict(os.environ)
if '$v'=='1': e['RNAGAN_LASTUP_WHOLE']='1'
print(subprocess.run([sys.executable,'bench.py','--precision','fp32','--steps','3','--warmup','6','--no-cpu-baseline','--no-roofline','--no-extras'],env=e,capture_output=True,text=True).stdout.strip().splitlines()[-1])" | grep -o 'ms_per_step": [0-9.]*')
  echo "whole=$v: fp32 $ms"
done; done
