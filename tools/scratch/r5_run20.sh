#!/bin/bash
# per-kernel durations of the fp32 iteration under f32mma = 1 (f32 MFMA) and 2 (six bf16 products)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_20; mkdir -p $O
for v in 1 2; do
  export RNAGAN_F32MMA=$v
  rm -rf $O/prof$v
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$v -- python3 bench.py --gpus 1 --precision fp32 --steps 4 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $O/bench$v.json 2> $O/bench$v.err
  python3 tools/prof_groups.py $O/prof$v "gemm" 40 > $O/groups$v.txt 2>&1
  python3 tools/prof_summary.py $O/prof$v 12 > $O/sum$v.txt 2>&1
  rm -rf $O/prof$v
done
head -20 $O/groups1.txt | cut -c1-230; head -20 $O/groups2.txt | cut -c1-230
