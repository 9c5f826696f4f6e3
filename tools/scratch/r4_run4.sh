#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_run4_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run4_tests.log
tail -8 gpurun_out/r4_run4_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_run4_bench.json 2> gpurun_out/r4_run4_bench.err
grep -E "timed region|extra fp32|FAILED" gpurun_out/r4_run4_bench.err | cut -c1-1500
for mode in 1 2 1 2; do
  RNAGAN_FORCE_DP=1 RNAGAN_DP_PREFIX_BWD=$mode python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_run4_dp$mode.json 2> gpurun_out/r4_run4_dp$mode.err
  echo "FORCE_DP prefix mode $mode: $(grep -E 'timed region' gpurun_out/r4_run4_dp$mode.err)"
done
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_run4_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r4_run4_smoke.log
