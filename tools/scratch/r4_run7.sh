#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_run7_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_run7_tests.log
tail -4 gpurun_out/r4_run7_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_run7_bench.json 2> gpurun_out/r4_run7_bench.err
grep -E "timed region|extra fp32|FAILED" gpurun_out/r4_run7_bench.err | cut -c1-1500
export MASTER_ADDR=127.0.0.1
p=29700
for r in 1 2; do
  python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2> gpurun_out/r4_dp_single.err | python3 -c "import json,sys; print('single', json.loads(sys.stdin.read())['ms_per_step'])"
  for mode in 1 2; do
    p=$((p+1))
    MASTER_PORT=$p RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 RNAGAN_DP_PREFIX_BWD=$mode python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2> gpurun_out/r4_dp_mode$mode.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_dp prefix mode $mode', d['ms_per_step'], d['config']['collective_backend'])"
  done
done
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_summary.py gpurun_out/r4_fp32_prof 40 > gpurun_out/r4_fp32_prof7.txt 2>&1
head -24 gpurun_out/r4_fp32_prof7.txt | cut -c1-200
rm -rf gpurun_out/r4_fp32_prof
