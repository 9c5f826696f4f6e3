#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/vae_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vae_prof -- python3 tools/bench_vae.py bf16 64 > gpurun_out/vae_prof.log 2>&1
cp $(ls gpurun_out/vae_prof/*/*kernel_stats.csv | head -1) gpurun_out/vae_kernel_stats.csv
rm -rf gpurun_out/vae_prof
tail -2 gpurun_out/vae_prof.log
