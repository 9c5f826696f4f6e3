#!/bin/bash
# round 5 evidence: full GPU suite, the driver's bench command, its kernel statistics, PMC passes, the package A/B against round 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final4; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err | cut -c1-300
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof.json 2> $O/prof.err
python3 tools/prof_summary.py $O/prof 70 > $O/prof.txt; head -12 $O/prof.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 tools/prof_groups.py $O/prof "" 90 > $O/prof_groups.txt 2>&1
rm -rf $O/prof
bash tools/pmc_bench.sh r5_final4/pmc > $O/pmc.log 2>&1; tail -30 $O/pmc.log | cut -c1-220
timeout 1500 python tools/ab_trees.py --trees r4=tools/scratch/_r4tree,head=. --rounds 5 --steps 40 --json $O/ab_trees.json > $O/ab_trees.log 2>&1; tail -7 $O/ab_trees.log | cut -c1-400
