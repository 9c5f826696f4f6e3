#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -q -x -k "g0_and_head or f32_matrix" 2>&1 | tail -3
python3 -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py -q -x -k "fp32 or float32 or reference_trainops or public_functional or tight or l2" 2>&1 | tail -3
for r in 1 2; do
  ms=$(python3 bench.py --precision fp32 --steps 3 --warmup 6 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "HEAD: fp32 $ms"
done
rm -rf gpurun_out/r4_fp32_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_fp32_prof -- python3 bench.py --gpus 1 --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4_fp32_prof.json 2> gpurun_out/r4_fp32_prof.err
python3 tools/prof_groups.py gpurun_out/r4_fp32_prof "" 60 > gpurun_out/r4_fp32_groups.txt 2>&1
rm -rf gpurun_out/r4_fp32_prof
grep -E "G0|Lin|last_up" gpurun_out/r4_fp32_groups.txt | cut -c1-200
