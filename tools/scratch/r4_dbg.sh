#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
echo "=== HEAD"; python3 -m pytest tests/test_ops_gpu.py -q -x -k "conv_down_up_wgrad or f32_matrix" 2>&1 | grep -E "rel err|passed|failed|Error" | head
cp tools/scratch/lib_nonarrow.so rna_gan_amd/librnagan_hip.so
echo "=== nonarrow"; python3 -m pytest tests/test_ops_gpu.py -q -x -k "conv_down_up_wgrad or f32_matrix" 2>&1 | grep -E "rel err|passed|failed|Error" | head
for r in 1 2; do for v in nonarrow HEAD; do
  if [ $v == HEAD ]; then cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so; else cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so; fi
  ms=$(python3 bench.py --precision fp32 --steps 3 --warmup 6 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "$v: fp32 $ms"
done; done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
