#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for v in HEAD cd_nodma cd_nomfma cd_neither; do
  if [ $v == HEAD ]; then cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so; else cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so; fi
  echo "== $v"; python3 tools/ab_conv.py --batch 64 --layers 0 --kinds down --sets "convd=1" --rounds 5 --check 0 2>&1 | grep "^L1" | cut -c1-110
done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
