#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_29; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "one_launch" > $O/pytest_op.log 2>&1; tail -4 $O/pytest_op.log | cut -c1-250
timeout 900 python tools/ab_step.py --variants "t256:wgrad_adam_tile=256;t128:wgrad_adam_tile=128" --rounds 5 --steps 40 --json $O/ab_tile.json > $O/ab.log 2>&1; tail -5 $O/ab.log | cut -c1-250
for v in 256 128; do
  export RNAGAN_WGRAD_ADAM_TILE=$v
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/prof$v.json 2> $O/prof$v.err
  python3 tools/prof_summary.py $O/prof 30 > $O/prof$v.txt
  rm -rf $O/prof
  echo "== tile $v"; head -1 $O/prof$v.txt; grep -n "wgrad8_kernel<true>\|wgrad_dma_kernel<false, true>\|adam_segs" $O/prof$v.txt | cut -c1-170
done
