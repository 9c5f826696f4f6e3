#!/bin/bash
# A/B of library builds on one box: image-side microbenchmark + step time + the image-side GPU tests on the variant
cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for r in 1 2 3; do for v in "$@"; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  sk=$(python tools/bench_skinny.py 2>/dev/null | awk '{print $1, $2}' | tr '\n' ' ')
  ms=$(python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "$v: $sk  bench $ms"
done; done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
