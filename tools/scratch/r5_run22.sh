#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_22; mkdir -p $O
export RNAGAN_F32MMA=2
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log | cut -c1-250
rm -rf $O/prof2
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python3 bench.py --gpus 1 --precision fp32 --steps 4 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $O/bench2.json 2> $O/bench2.err
python3 tools/prof_groups.py $O/prof2 "gemm" 40 > $O/groups2.txt 2>&1
rm -rf $O/prof2
head -24 $O/groups2.txt | cut -c1-160
timeout 600 python3 bench.py --gpus 1 --precision fp32 --steps 6 --warmup 12 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*'
