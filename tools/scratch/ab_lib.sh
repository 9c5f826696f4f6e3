#!/bin/bash
# A/B of two builds of the library on one box: tools/scratch/lib_old.so vs lib_new.so (copied over rna_gan_amd/librnagan_hip.so)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in old new; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  echo "$v: $(python tools/ab_conv.py --check 0 --rounds 5 --layers 0 --kinds up --sets "convp=1" 2>&1 | grep "^L1" | cut -c1-90)  bench $(python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')"
done; done
cp tools/scratch/lib_new.so rna_gan_amd/librnagan_hip.so
python -m pytest tests/test_ops_gpu.py -x -q -k "convp or sign_bits" 2>&1 | tail -2
