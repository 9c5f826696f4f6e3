#!/bin/bash
# A/B of library builds on one box: step time of the benchmark, 3 interleaved rounds:  ab_step.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for r in 1 2 3; do for v in "$@"; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  ms=$(python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "$v: bench $ms"
done; done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
