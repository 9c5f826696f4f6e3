#!/bin/bash
# probe builds of convd_kernel (device durations from the kernel trace, not host-paired timing):
# a = no DMA, no LDS reads; b = + no barrier; c = no DMA, no barrier; d = no DMA; e = no MFMA; f = no MFMA, no DMA
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for v in HEAD cd_a cd_b cd_c cd_d cd_nomfma cd_neither; do
  if [ $v == HEAD ]; then cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so; else cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so; fi
  rm -rf gpurun_out/cdp
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cdp -- python3 tools/ab_conv.py --batch 64 --layers 0 --kinds down --sets "convd=1" --rounds 3 --check 0 > /dev/null 2>&1
  echo "== $v $(python3 tools/prof_groups.py gpurun_out/cdp convd 3 | grep convd | cut -c1-60)"
done
rm -rf gpurun_out/cdp
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
