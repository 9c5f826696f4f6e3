#!/bin/bash
# round 5, call 3: the full GPU suite at HEAD + the driver-style bench line (with the new extras) + kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_3
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_3/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_3/pytest.log
tail -5 gpurun_out/r5_3/pytest.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_3/bench.json 2> gpurun_out/r5_3/bench.err
tail -25 gpurun_out/r5_3/bench.err
rm -rf gpurun_out/r5_3/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_3/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5_3/prof.json 2> gpurun_out/r5_3/prof.err
python3 tools/prof_summary.py gpurun_out/r5_3/prof 70 > gpurun_out/r5_3/prof.txt
head -14 gpurun_out/r5_3/prof.txt
cp $(ls gpurun_out/r5_3/prof/*/*kernel_stats.csv | head -1) gpurun_out/r5_3/kernel_stats.csv
python3 tools/prof_groups.py gpurun_out/r5_3/prof "" 80 > gpurun_out/r5_3/prof_groups.txt 2>&1
rm -rf gpurun_out/r5_3/prof
