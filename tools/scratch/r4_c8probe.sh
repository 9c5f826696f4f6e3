#!/bin/bash
# what bounds the 512 x 128-tile conv8 launches (layer 1 down, layer 2 up)?  probe builds: A gather folded into a 2 MB window
# (all L2 hits), no epilogue stores, both -- results of the probes are wrong by construction, only the times count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for r in 1 2; do for v in c8base c8fold c8noepi c8both; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  echo "== $v (round $r)"
  python3 tools/ab_conv.py --batch 64 --layers 0,1,2 --kinds down,up --sets "conv8=5" --check 0 --rounds 5 2>/dev/null | grep -E "^L"
done; done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
