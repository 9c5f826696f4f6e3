#!/bin/bash
# round-4 evidence: GPU suite, smoke, the driver's bench command, its rocprofv3 kernel statistics, the PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_fin2_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_fin2_tests.log
tail -4 gpurun_out/r4_fin2_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_fin2_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r4_fin2_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_fin2_bench.json 2> gpurun_out/r4_fin2_bench.err
grep -E "timed region|extra|cpu baseline:" gpurun_out/r4_fin2_bench.err | cut -c1-1400
rm -rf gpurun_out/r4_fin2_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fin2_prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r4_fin2_prof.json 2> gpurun_out/r4_fin2_prof.err
python3 tools/prof_summary.py gpurun_out/r4_fin2_prof 60 > gpurun_out/r4_fin2_prof.txt
head -12 gpurun_out/r4_fin2_prof.txt
cp $(ls gpurun_out/r4_fin2_prof/*/*kernel_stats.csv | head -1) gpurun_out/r4_fin2_kernel_stats.csv
rm -rf gpurun_out/r4_fin2_prof
bash tools/pmc_bench.sh r4pmc2 > gpurun_out/r4_pmc2.log 2>&1; tail -12 gpurun_out/r4_pmc.log | cut -c1-300
