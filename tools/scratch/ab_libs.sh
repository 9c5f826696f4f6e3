#!/bin/bash
# A/B of several BUILDS of the library on one box: ab_libs.sh name1 name2 ...  (tools/scratch/lib_<name>.so), interleaved rounds:
# conv microbench totals (down / up over the five layers) and the benchmark's step time
cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for r in 1 2 3; do for v in "$@"; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  conv=$(python tools/ab_conv.py --check 0 --rounds 5 --sets "conv8=5" 2>&1 | grep "set0: total" | awk '{print $1, $4, $6}' | tr '\n' ' ')
  ms=$(python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "$v: $conv  bench $ms"
done; done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
