#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
for v in cd_pp HEAD; do
  if [ $v == HEAD ]; then cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so; else cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so; fi
  echo "== $v"; timeout 300 python3 -m pytest tests/test_ops_gpu.py -q -x -k "convd_plane" 2>&1 | tail -2
  rm -rf gpurun_out/cdp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cdp -- python3 tools/ab_conv.py --batch 64 --layers 0 --kinds down --sets "convd=1" --rounds 3 --check 0 > /dev/null 2>&1
  python3 tools/prof_groups.py gpurun_out/cdp convd 3 | grep convd | cut -c1-70
done
rm -rf gpurun_out/cdp
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
