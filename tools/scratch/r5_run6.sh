#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_6; mkdir -p $O
for v in A B C; do
  cp tools/scratch/lib_cpdbg$v.so rna_gan_amd/librnagan_hip.so
  echo "== $v"; timeout 200 python tools/scratch/dbg_convp64.py 2>&1 | grep "mismatch"
done > $O/dbg.log 2>&1
cat $O/dbg.log
