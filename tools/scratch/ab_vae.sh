#!/bin/bash
# A/B of library builds on one box: betaVAE training iteration (tools/bench_vae.py), 3 interleaved rounds
cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for r in 1 2 3; do for v in "$@"; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  echo "$v: $(python tools/bench_vae.py bf16 64 2>/dev/null | tail -1 | cut -c1-40)"
done; done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
