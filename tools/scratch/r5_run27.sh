#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_27; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py tests/test_bench_step_gpu.py tests/test_quality_ab_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; tail -4 $O/pytest_train.log | cut -c1-250
timeout 900 python tools/ab_step.py --variants "on:losses.SKINNY_SLAB_ADAM=1;off:losses.SKINNY_SLAB_ADAM=0" --rounds 5 --steps 40 --json $O/ab_skinny_slab.json > $O/ab.log 2>&1; tail -5 $O/ab.log | cut -c1-250
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof.json 2> $O/prof.err
python3 tools/prof_summary.py $O/prof 70 > $O/prof.txt; head -10 $O/prof.txt
python3 tools/prof_groups.py $O/prof "" 90 > $O/prof_groups.txt 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/prof
grep -n "reduce_slabs_wide\|skinny_wgrad" $O/prof_groups.txt | cut -c1-160
tail -1 $O/prof.json | cut -c1-400
