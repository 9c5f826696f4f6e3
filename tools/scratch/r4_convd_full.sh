#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_cdfull_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_cdfull_tests.log
tail -3 gpurun_out/r4_cdfull_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for r in 1 2 3; do for v in 0 1; do
  ms=$(python3 -c "
import os,subprocess,sys
e=dict(os.environ); e['RNAGAN_CONVD']='$v'
print(subprocess.run([sys.executable,'bench.py','--steps','20','--warmup','5','--no-cpu-baseline','--no-roofline','--no-extras'],env=e,capture_output=True,text=True).stdout.strip().splitlines()[-1])" | grep -o 'ms_per_step": [0-9.]*')
  echo "convd=$v: $ms"
done; done
