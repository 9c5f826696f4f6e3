#!/bin/bash
# mid-round check at HEAD: full GPU suite, smoke, default bench, fp32 kernel stats + fp32 PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_mid_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_mid_tests.log
tail -4 gpurun_out/r4_mid_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_mid_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r4_mid_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_mid_bench.json 2> gpurun_out/r4_mid_bench.err
grep -E "timed region|extra|cpu baseline:" gpurun_out/r4_mid_bench.err | cut -c1-1400
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_summary.py gpurun_out/r4_fp32_prof 40 > gpurun_out/r4_fp32_kernel_summary.txt 2>&1
python3 tools/prof_groups.py gpurun_out/r4_fp32_prof "" 60 > gpurun_out/r4_fp32_groups.txt 2>&1
rm -rf gpurun_out/r4_fp32_prof
bash tools/scratch/r4_f32pmc.sh > gpurun_out/r4_f32pmc.log 2>&1; tail -30 gpurun_out/r4_f32pmc.log | cut -c1-250
