#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_run5_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run5_tests.log
tail -8 gpurun_out/r4_run5_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_run5_bench.json 2> gpurun_out/r4_run5_bench.err
grep -E "timed region|extra fp32|FAILED" gpurun_out/r4_run5_bench.err | cut -c1-1500
for mode in 1 2 1 2; do
  RNAGAN_FORCE_DP=1 RNAGAN_DP_PREFIX_BWD=$mode python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_run5_dp$mode.json 2> gpurun_out/r4_run5_dp$mode.err
  echo "FORCE_DP prefix mode $mode: $(grep -E 'timed region' gpurun_out/r4_run5_dp$mode.err)"
done
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_summary.py gpurun_out/r4_fp32_prof 40 > gpurun_out/r4_fp32_prof.txt 2>&1
head -45 gpurun_out/r4_fp32_prof.txt | cut -c1-200
rm -rf gpurun_out/r4_fp32_prof
