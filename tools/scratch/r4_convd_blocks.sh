#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 256 512 1024 256 512 1024; do
  rm -rf gpurun_out/cdp
  RNAGAN_CONVD_BLOCKS=$b rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cdp -- python3 tools/ab_conv.py --batch 64 --layers 0 --kinds down --sets "convd=1" --rounds 3 --check 0 > /dev/null 2>&1
  echo "== blocks $b $(python3 tools/prof_groups.py gpurun_out/cdp convd 3 | grep convd | cut -c1-75)"
done
rm -rf gpurun_out/cdp
