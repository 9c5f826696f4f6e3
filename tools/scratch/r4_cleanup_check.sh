#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -q -x -k "conv_down_up_wgrad or f32_matrix or g0_and_head or linear or image_side or upconv3" 2>&1 | tail -3
python3 -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py tests/test_vae_gpu.py -q -x -k "fp32 or float32 or reference_trainops or public_functional or tight or l2 or vae" 2>&1 | tail -3
for r in 1 2; do
  ms=$(python3 bench.py --precision fp32 --steps 3 --warmup 6 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step": [0-9.]*')
  echo "HEAD: fp32 $ms"
done
